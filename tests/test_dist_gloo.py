"""CPU, world_size 2, gloo: the multi-GPU path (voltrix.dist.RowShardedSpMM) -- row-window partition, column remap,
padded all-gather of B, per-rank SpMM on the shard -- reproduces the single-process result.  The local compute is
injected from here (the oracle's CPU block-format SpMM); on the GPU box the same class runs the HIP path
(tests/test_gpu_dist.py, bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, PKG_ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_shard(rank, world, port, num_feats, scale, out_dir, mode, slabs, graph="reddit_like"):
    """The round-3 construction: every rank builds ONLY its own shard (synth_graphs rows=...), partitions from the shared
    degree sequence, and exchanges B with the chosen schedule."""
    for p in (REPO, PKG_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth_graphs
        from oracle import oracle_c
        from voltrix.dist import RowShardedSpMM, partition_rows

        deg = synth_graphs.target_degrees(graph, scale=scale)
        n = deg.numel()
        full_indptr = torch.zeros(n + 1, dtype=torch.int64)
        full_indptr[1:] = torch.cumsum(deg, 0)
        parts = partition_rows(full_indptr, n, world)
        r0, r1 = parts[rank]
        local_indptr, local_indices, _ = synth_graphs.generate(graph, scale=scale, rows=(r0, r1))
        gen = torch.Generator().manual_seed(5)
        feat = torch.randn(n, num_feats, generator=gen)
        op = RowShardedSpMM.from_shard(
            local_indptr, local_indices, n, parts, mode=mode, slabs=slabs,
            local_preprocess=lambda ip, ix, rows: oracle_c.csr_preprocess(ip.numpy(), ix.numpy(), rows),
            local_spmm=lambda h, rows, e, b: torch.from_numpy(oracle_c.spmm_blocked(h[0], h[1], h[2], rows, b.numpy(), "none")))
        out = op(feat[r0:r1].contiguous())
        assert out.shape == (r1 - r0, num_feats)
        assert torch.equal(out, op(feat[r0:r1].contiguous()))
        if mode == "rows":      # fewer rows travel than the all-gather moves, and only referenced ones
            assert op.compact_rows <= (r1 - r0) + sum(p[1] - p[0] for i, p in enumerate(parts) if i != rank)
            assert op.exchange_bytes_received(num_feats, 4) == sum(op._want) * num_feats * 4
        np.save(os.path.join(out_dir, f"out_{rank}.npy"), out.numpy())
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), np.array([r0, r1]))
        np.save(os.path.join(out_dir, f"csr_{rank}.npy"), np.concatenate([local_indptr.numpy(), local_indices.numpy()]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,slabs", [("collective", 1), ("p2p", 1), ("collective", 3), ("p2p", 2), ("rows", 1)])
def test_row_sharded_from_own_shard_world2(tmp_path, mode, slabs):
    from oracle import oracle_c

    world, num_feats, scale = 2, 40, 0.004
    mp.spawn(_worker_shard, args=(world, _free_port(), num_feats, scale, str(tmp_path), mode, slabs), nprocs=world, join=True)
    gen = torch.Generator().manual_seed(5)
    feat = None
    covered = 0
    for r in range(world):
        r0, r1 = np.load(tmp_path / f"rows_{r}.npy")
        csr = np.load(tmp_path / f"csr_{r}.npy")
        ip, ix = csr[: r1 - r0 + 1], csr[r1 - r0 + 1:]
        if feat is None:
            n = int(ix.max()) + 1 if False else None
        covered += r1 - r0
        assert r0 % 16 == 0
    n = covered
    feat = torch.randn(n, num_feats, generator=gen)
    for r in range(world):
        r0, r1 = np.load(tmp_path / f"rows_{r}.npy")
        csr = np.load(tmp_path / f"csr_{r}.npy")
        ip, ix = csr[: r1 - r0 + 1].astype(np.int64), csr[r1 - r0 + 1:].astype(np.int64)
        ref = np.zeros((r1 - r0, num_feats), np.float64)       # the shard's rows of csr(ones) @ feat, global column ids
        rows = np.repeat(np.arange(r1 - r0), np.diff(ip))
        np.add.at(ref, rows, feat.numpy().astype(np.float64)[ix])
        got = np.load(tmp_path / f"out_{r}.npy")
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-6


def test_referenced_rows_exchange_moves_only_what_the_shards_reference(tmp_path):
    """world 3, a sparse graph (cora-like: 3.9 edges per row, so every shard references a FRACTION of the others' rows): the
    referenced-rows exchange gives every rank the product of its own shard and receives fewer rows than an all-gather moves."""
    world, num_feats = 3, 16
    mp.spawn(_worker_shard, args=(world, _free_port(), num_feats, 1.0, str(tmp_path), "rows", 1, "cora_like"), nprocs=world,
             join=True)
    rows = [tuple(np.load(tmp_path / f"rows_{r}.npy")) for r in range(world)]
    n = rows[-1][1]
    feat = torch.randn(n, num_feats, generator=torch.Generator().manual_seed(5))
    for r, (r0, r1) in enumerate(rows):
        csr = np.load(tmp_path / f"csr_{r}.npy")
        ip, ix = csr[: r1 - r0 + 1].astype(np.int64), csr[r1 - r0 + 1:].astype(np.int64)
        ref = np.zeros((r1 - r0, num_feats), np.float64)
        np.add.at(ref, np.repeat(np.arange(r1 - r0), np.diff(ip)), feat.numpy().astype(np.float64)[ix])
        got = np.load(tmp_path / f"out_{r}.npy")
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-6
        remote_refs = len(np.unique(ix[(ix < r0) | (ix >= r1)]))
        assert 0 < remote_refs < n - (r1 - r0)                       # a strict subset of the other shards' rows


def _worker_choose(rank, world, port, out_dir):
    """choose_exchange: every rank times the candidates with ITS OWN clock (rank 1's is skewed so that the ranks disagree on
    which schedule was faster locally), the all-reduced MAX must make both keep the same one; own_rows writes B in place."""
    for p in (REPO, PKG_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth_graphs
        from oracle import oracle_c
        from voltrix.dist import RowShardedSpMM

        indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.004)
        n = indptr.numel() - 1
        feat = torch.randn(n, 24, generator=torch.Generator().manual_seed(5))
        op = RowShardedSpMM(indptr, indices, n,
                            local_preprocess=lambda ip, ix, rows: oracle_c.csr_preprocess(ip.numpy(), ix.numpy(), rows),
                            local_spmm=lambda h, rows, e, b: torch.from_numpy(
                                oracle_c.spmm_blocked(h[0], h[1], h[2], rows, b.numpy(), "none")))
        ref = op(feat[op.row_start:op.row_end].contiguous())
        # fake clock: consecutive readings; candidate k's interval on this rank = table[rank][k] seconds
        table = {0: [0.010, 0.030], 1: [0.050, 0.020]}[rank]    # rank 0 alone would pick collective, rank 1 p2p
        ticks = []
        for k in range(2):
            ticks += [100.0 * k, 100.0 * k + table[k] * 2]      # iters = 2
        it = iter(ticks)
        timings = op.choose_exchange(feat[op.row_start:op.row_end], modes=("collective", "p2p"), iters=2, clock=lambda: next(it))
        # MAX over ranks: collective 50 ms, p2p 30 ms -> p2p on BOTH ranks
        assert abs(timings["collective"] - 50.0) < 1e-6 and abs(timings["p2p"] - 30.0) < 1e-6, timings
        assert op.mode == "p2p"
        mine = op.own_rows(24, feat)                             # the producer writes B straight into the gather buffer
        mine.copy_(feat[op.row_start:op.row_end])
        out = op.multiply(op.gather_into(op._buffer("whole", 24, feat), mine))
        assert torch.equal(out, ref)
        np.save(os.path.join(out_dir, f"mode_{rank}.npy"), np.array([0 if op.mode == "collective" else 1]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_exchange_schedule_is_chosen_from_measurement_and_agreed_on_by_all_ranks(tmp_path):
    mp.spawn(_worker_choose, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / "mode_0.npy")[0]) == int(np.load(tmp_path / "mode_1.npy")[0]) == 1


def test_partition_is_the_same_on_any_device_and_predictions_are_sane():
    import synth_graphs
    from voltrix.dist import partition_rows, predicted_step_ms

    deg = synth_graphs.target_degrees("products_like", scale=0.01)
    indptr = torch.zeros(deg.numel() + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(deg, 0)
    for world in (1, 2, 3, 8):
        parts = partition_rows(indptr, deg.numel(), world)
        assert parts[0][0] == 0 and parts[-1][1] == deg.numel() and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        edges = [int(indptr[b] - indptr[a]) for a, b in parts]
        assert max(edges) - min(edges) <= 2 * 16 * int(deg.max())      # balanced to the window granularity
    pred = predicted_step_ms(8, 3.55e9, 8.0)
    assert 20 < pred["allgather_direct_ms"] < 27 and 150 < pred["allgather_ring_ms"] < 240
    assert pred["step_direct_overlapped_ms"] == max(pred["allgather_direct_ms"], 8.0)
    assert predicted_step_ms(1, 3.55e9, 62.0)["step_direct_ms"] == 62.0


def _worker(rank, world, port, num_feats, scale, out_dir):
    for p in (REPO, PKG_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth_graphs
        from oracle import oracle_c
        from voltrix.dist import RowShardedSpMM

        indptr, indices, _ = synth_graphs.generate("reddit_like", scale=scale)
        n = indptr.numel() - 1
        gen = torch.Generator().manual_seed(5)
        feat = torch.randn(n, num_feats, generator=gen)

        def local_preprocess(ip, ix, rows):
            return oracle_c.csr_preprocess(ip.numpy(), ix.numpy(), rows)

        def local_spmm(handle, rows, edges, b):
            return torch.from_numpy(oracle_c.spmm_blocked(handle[0], handle[1], handle[2], rows, b.numpy(), "none"))

        op = RowShardedSpMM(indptr, indices, n, local_preprocess=local_preprocess, local_spmm=local_spmm)
        assert op.world_size == world and op.rank == rank
        out = op(feat[op.row_start:op.row_end].contiguous())
        assert out.shape == (op.local_rows, num_feats)
        out2 = op(feat[op.row_start:op.row_end].contiguous())  # gather buffer reuse
        assert torch.equal(out, out2)
        np.save(os.path.join(out_dir, f"out_{rank}.npy"), out.numpy())
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), np.array([op.row_start, op.row_end]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_row_sharded_spmm_world2(tmp_path, world):
    import synth_graphs
    from oracle import oracle_c

    num_feats, scale = 24, 0.004
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_feats, scale, str(tmp_path)), nprocs=world, join=True)

    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=scale)
    n = indptr.numel() - 1
    gen = torch.Generator().manual_seed(5)
    feat = torch.randn(n, num_feats, generator=gen)
    ref = oracle_c.spmm_csr(indptr.numpy(), indices.numpy(), feat.numpy(), n)
    got = np.zeros_like(ref)
    covered = 0
    for r in range(world):
        r0, r1 = np.load(tmp_path / f"rows_{r}.npy")
        got[r0:r1] = np.load(tmp_path / f"out_{r}.npy")
        covered += r1 - r0
        assert r0 % 16 == 0
    assert covered == n
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-6


def test_single_process_path_is_identity_remap():
    import synth_graphs
    from oracle import oracle_c
    from voltrix.dist import RowShardedSpMM

    indptr, indices, _ = synth_graphs.generate("cora_like")
    n = indptr.numel() - 1
    feat = torch.randn(n, 8)
    op = RowShardedSpMM(indptr, indices, n,
                        local_preprocess=lambda ip, ix, rows: oracle_c.csr_preprocess(ip.numpy(), ix.numpy(), rows),
                        local_spmm=lambda h, rows, e, b: torch.from_numpy(
                            oracle_c.spmm_blocked(h[0], h[1], h[2], rows, b.numpy(), "none")))
    assert op.parts == [(0, n)] and op.rows_padded == n
    ref = oracle_c.spmm_csr(indptr.numpy(), indices.numpy(), feat.numpy(), n)
    assert np.allclose(op(feat).numpy(), ref, rtol=1e-5, atol=1e-5)
