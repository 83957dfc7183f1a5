"""Child process of tests/test_gpu_dist.py::test_row_sharded_world2_two_processes_hip_path: one rank of a world-size-2
run of voltrix.dist.RowShardedSpMM on the HIP path (both ranks on cuda:0, gloo for the all-gather of B), checked against
the oracle (torch.sparse.mm on the CPU = the reference's own oracle call) on the rank's rows.

    python tests/dist_gpu_worker.py RANK WORLD PORT OUT_JSON [two-level]
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "voltrix-spmm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(REPO, "voltrix-spmm_amd", ".jit_cache"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    two_level = len(sys.argv) > 5 and sys.argv[5] == "two-level"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, VOLTRIX_TUNE_SPACE="none",
                      VOLTRIX_HYBRID="1" if two_level else "0", VOLTRIX_HYBRID_MIN_SHARE="0")
    import torch
    import torch.distributed as dist

    import synth_graphs
    import voltrix
    from oracle import torch_ref
    from voltrix.dist import RowShardedSpMM

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    result = {"rank": rank, "ok": False}
    try:
        indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)     # same seed on every rank
        n = indptr.numel() - 1
        num_feats = 96
        gen = torch.Generator().manual_seed(11)
        feat = torch.randn(n, num_feats, generator=gen).half()
        op = RowShardedSpMM(indptr, indices, n, device=torch.device("cuda", 0), hash_tag=f"dist2p_{two_level}")
        assert op.world_size == world and op.rank == rank and op.row_start % 16 == 0
        # the shard is rectangular: local rows x (world * rows_padded) remapped columns
        assert world * op.rows_padded >= n and op.local_rows < n
        hint = voltrix.two_level_of(op.handle[1])
        if two_level:
            assert hint is not None and hint.plan.num_ksteps > 0 and hint.num_nodes == op.local_rows
        else:
            assert hint is None
        out = op(feat[op.row_start:op.row_end].cuda())
        out2 = op(feat[op.row_start:op.row_end].cuda())
        torch.cuda.synchronize()
        assert out.is_cuda and out.shape == (op.local_rows, num_feats) and torch.equal(out, out2)
        ref = torch_ref.spmm(indptr, indices, feat.float(), n)[op.row_start:op.row_end]
        rel = float((out.cpu() - ref).norm() / ref.norm())
        deg = (indptr[1:] - indptr[:-1])[op.row_start:op.row_end].double()
        aabs = torch_ref.spmm(indptr, indices, feat.float().abs(), n)[op.row_start:op.row_end].double()
        bound = (deg[:, None] + 1) * 2.0 ** -23 * aabs + 1e-30     # same operand, accumulation order only
        worst = float(((out.cpu().double() - ref.double()).abs() / bound).max())
        result.update(ok=bool(rel < 1e-6 and worst <= 1.0), rel=rel, worst_bound_ratio=worst, rows=[op.row_start, op.row_end],
                      edges=op.local_edges, two_level=bool(hint is not None),
                      shared_edges=int(hint.plan.num_shared_edges) if hint is not None else 0)
        dist.barrier()
    finally:
        with open(out_path, "w") as f:
            json.dump(result, f)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
