#!/usr/bin/env python3
"""Headline benchmark: SpMM GFLOP/s + achieved HBM GB/s, reddit-like CSR x dense feat=128 fp16 (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole synthetic graph: (N > 1: RCCL all-gather of the dense operand B,
then) the tiled SpMM accumulate on every rank's row-window shard -- in the reference's window format, or in the two-level
format (shared columns of 512-row panels on the panel kernel beside the window kernel); the sweep
before the timed region times both and keeps the faster (--format window|two-level forces one).  Inputs are resident in HBM before the timed region;
preprocessing (CSR -> block format) is done once, outside it, as in the reference's protocol (bench/bm_voltrix.py:17,36).
Rank 0 prints ONE JSON line.  N > 1 shards the SAME matrix by row windows (strong scaling).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.join(REPO, "voltrix-spmm_amd")
for _p in (REPO, PKG_ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG_ROOT, ".jit_cache"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import synth_graphs  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured-achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="reddit_like", choices=sorted(synth_graphs.CONFIGS))
    ap.add_argument("--feat", type=int, default=None, help="feature width (default: the workload's)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the node count (debug only)")
    ap.add_argument("--tile", default=None, help="fs,depth,waves[,sched] (default: quick sweep over the tile space)")
    ap.add_argument("--format", default="auto", choices=["auto", "window", "two-level"],
                    help="window: the reference's block format only; two-level: shared columns of 512-row panels on the "
                         "panel kernel + the rest in the block format (voltrix/hybrid.py); auto: time both, keep the faster")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for debugging)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: run all-gather and SpMM back to back on one stream (default: the all-gather of step k+1 "
                         "runs on a second stream beside the SpMM of step k, double-buffered B)")
    ap.add_argument("--one-device", action="store_true",
                    help="debug: every rank uses cuda:0 (exercises the sharded path on a 1-GPU box, with --backend gloo)")
    return ap.parse_args()


def cpu_baseline(indptr, indices, num_nodes, num_feats, seed=0):
    """torch.sparse.mm (the reference's own oracle call, tests/test_spmm.py:24-29) on the host cores, fp32 -- CPU fp16
    CSR mm is not implemented in torch.  Bounded sample: the whole matrix up to 150 M edges, else a contiguous 1/16 row
    sample."""
    from oracle import torch_ref  # checker / baseline leg only

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    ip, ix = indptr.cpu(), indices.cpu()
    rows, sample = num_nodes, "all rows"
    if ix.numel() > 150_000_000:
        rows = (num_nodes // 16) // 16 * 16
        sample = f"contiguous 1/16 row sample (rows 0..{rows})"
        ip = ip[: rows + 1].clone()
        ix = ix[: int(ip[-1])].clone()
    a = torch_ref.csr_ones(ip, ix, rows, num_nodes)
    gen = torch.Generator().manual_seed(seed)
    feat = torch.randn(num_nodes, num_feats, generator=gen)
    for _ in range(2):
        a @ feat
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        a @ feat
        times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    nnz = int(ix.numel())
    return {
        "value": 2.0 * nnz * num_feats / t / 1e9,
        "unit": "GFLOP/s",
        "cores": cores,
        "kind": "port",
        "sample": f"torch.sparse.mm(csr(ones fp32), feat fp32) on CPU, {sample}, nnz={nnz}, F={num_feats}, "
                  f"median of 5 after 2 warm-ups ({t * 1e3:.1f} ms); the reference's own CPU oracle call",
        "ms": t * 1e3,
    }


def vendor_baseline(indptr, indices, num_nodes, num_feats, device):
    """The reference's own GPU baseline (tests/test_spmm.py:61-72: `sparse.cuda() @ feat.cuda()` = cuSPARSE there):
    torch.sparse.mm on the GPU = hipSPARSE/rocSPARSE SpMM, fp32 (the fp16 CSR SpMM is not implemented in hipSPARSE).
    The reference publishes only speedups over this kind of baseline."""
    try:
        a = torch.sparse_csr_tensor(indptr, indices, torch.ones(indices.numel(), device=device), size=(num_nodes, num_nodes))
        feat = torch.randn(num_nodes, num_feats, device=device)
        for _ in range(2):
            a @ feat
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            a @ feat
        e.record()
        e.synchronize()
        return {"name": "torch.sparse.mm on the GPU (hipSPARSE CSR SpMM), fp32 values and features",
                "ms": s.elapsed_time(e) / 3}
    except Exception as exc:  # not implemented / out of memory: report, do not fail the bench
        return {"name": "torch.sparse.mm on the GPU (hipSPARSE CSR SpMM)", "error": str(exc)[:200]}


def measured_traffic(workload, num_feats, dtype):
    """Per-launch fabric-side bytes of the dominant kernel from the committed rocprofv3 PMC summary (profiles/), or
    None: the counters need their own profiling passes and cannot be read live in this process."""
    try:
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            entry = json.load(f).get(f"{workload}|{num_feats}|{dtype}")
        return None if entry is None else entry["traffic_bytes"]
    except (OSError, ValueError, KeyError):
        return None


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    import voltrix
    from voltrix import capi
    from voltrix import dist as vdist

    cfg = synth_graphs.CONFIGS[args.workload]
    num_feats = args.feat or cfg["feat"]
    is_f16 = args.dtype == "f16"
    in_bytes = 2 if is_f16 else 4

    # ---- synthetic graph (same seed on every rank), row-window shard, block-format handle ------------------------
    indptr, indices, _ = synth_graphs.generate(args.workload, device=device, scale=args.scale)
    num_nodes, nnz = indptr.numel() - 1, indices.numel()
    parts = vdist.partition_rows(indptr, num_nodes, world)
    r0, r1 = parts[rank]
    rows_padded = max(1, max(p[1] - p[0] for p in parts))
    local_indptr, local_indices = vdist.shard_csr(indptr, indices, num_nodes, parts, rank)
    if world > 1:
        local_indices = vdist.remap_columns(local_indices, parts, rows_padded)
    local_rows, local_nnz = r1 - r0, local_indices.numel()
    num_cols = world * rows_padded if world > 1 else num_nodes   # ids index the gathered B (padded shards) when sharded
    preprocess_ms = None
    for _ in range(2):   # first call pays library load / allocator warm-up; report the second (host wall clock, sync'd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        blk_offsets, hspa_packed, hind, _ = voltrix.csr_fused_preprocess_kernel(local_indptr, local_indices,
                                                                                local_rows, num_cols=num_cols)
        torch.cuda.synchronize()
        preprocess_ms = (time.perf_counter() - t0) * 1e3
    total_blocks = int(blk_offsets[-1])

    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    feat_local = torch.randn(local_rows, num_feats, generator=gen, device=device,
                             dtype=torch.float32).to(torch.float16 if is_f16 else torch.float32)
    if world > 1:
        gathered = torch.zeros(world * rows_padded, num_feats, dtype=feat_local.dtype, device=device)
        send = gathered[rank * rows_padded:(rank + 1) * rows_padded]
        send[:local_rows].copy_(feat_local)
    else:
        gathered, send = feat_local, None
    out = torch.empty(local_rows, num_feats, dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    ptrs = (blk_offsets.data_ptr(), hspa_packed.data_ptr(), hind.data_ptr())

    # "balance" schedule of this rank's handle (length-sorted windows inside 256-window chunks), computed on the GPU
    from voltrix.jit_kernels.spmm import ORDER_CHUNKS

    orders = {0: 0}
    order_keep = []
    for sched, chunk in ORDER_CHUNKS.items():
        o = torch.empty((local_rows + 15) // 16, dtype=torch.int32, device=device)
        capi.launch_window_order(blk_offsets, local_rows, o, stream, chunk)
        order_keep.append(o)
        orders[sched] = o.data_ptr()

    from voltrix import hybrid

    def window_launch(handle_ptrs, handle_nnz, order_ptrs, tile, b_ptr, dst):
        rc = capi.launch_spmm(handle_ptrs[0], handle_ptrs[1], handle_ptrs[2], local_rows, handle_nnz, num_feats, b_ptr,
                              dst.data_ptr(), is_f16, tile[:3], stream, order_ptrs[tile[3]])
        assert rc == 0, f"voltrix_launch_spmm rc={rc}"

    # ---- two-level format (voltrix/hybrid.py): plan + residual handle per (waves, row_blocks, tau), built on demand ----
    two_level_cache = {}
    side_stream = torch.cuda.Stream(device=device)
    main_stream = torch.cuda.current_stream()

    def two_level_state(waves, rb, tau):
        key = (waves, rb, tau)
        if key not in two_level_cache:
            for _ in range(2):   # report the warm build (host wall clock, sync'd): plan + residual handle
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r_indptr, r_indices, plan = hybrid.build_panel_plan(local_indptr, local_indices, local_rows, num_cols,
                                                                    waves, rb, tau)
                r_handle = voltrix.csr_fused_preprocess_kernel(r_indptr, r_indices, local_rows, num_cols=num_cols)
                torch.cuda.synchronize()
                build_ms = (time.perf_counter() - t0) * 1e3
            r_orders, keep = {0: 0}, []
            for sched, chunk in ORDER_CHUNKS.items():
                o = torch.empty((local_rows + 15) // 16, dtype=torch.int32, device=device)
                capi.launch_window_order(r_handle[0], local_rows, o, stream, chunk)
                keep.append(o)
                r_orders[sched] = o.data_ptr()
            two_level_cache[key] = dict(plan=plan, handle=r_handle, nnz=r_indices.numel(), orders=r_orders, keep=keep,
                                        shared=torch.empty(local_rows, num_feats, dtype=torch.float32, device=device),
                                        build_ms=build_ms, blocks=int(r_handle[0][-1]))
        return two_level_cache[key]

    def spmm(cand, b_full=None):
        """One SpMM in the candidate's format.  cand = ("window", fs, depth, waves, sched) or
        ("two-level", fs, depth, waves, sched, plan_waves, row_blocks, tau, panel_depth)."""
        b = gathered if b_full is None else b_full
        if cand[0] == "window":
            window_launch(ptrs, local_nnz, orders, cand[1:5], b.data_ptr(), out)
            return
        st = two_level_state(*cand[5:8])
        plan, h = st["plan"], st["handle"]
        if plan.num_ksteps == 0:
            window_launch((h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr()), st["nnz"], st["orders"], cand[1:5],
                          b.data_ptr(), out)
            return
        ptile = (min(128, cand[1]), cand[8], 1 if cand[1] >= 128 else 2)
        fork = torch.cuda.Event()
        fork.record(main_stream)
        side_stream.wait_event(fork)
        hybrid.launch_panel(plan, b, st["shared"], accumulate=False, tile=ptile, stream=side_stream.cuda_stream)
        join = torch.cuda.Event()
        join.record(side_stream)
        window_launch((h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr()), st["nnz"], st["orders"], cand[1:5], b.data_ptr(),
                      out)
        main_stream.wait_event(join)
        capi.launch_add_inplace_f32(out, st["shared"], stream)

    # ---- candidate: explicit, or a quick sweep over the instantiated space (what the autotuner does on first call) ----
    two_level_ok = is_f16 and args.format != "window"
    if args.tile:
        tile = tuple(int(x) for x in args.tile.split(","))
        tile = tile if len(tile) == 4 else tile + (1,)
        cands = [("window",) + tile] if args.format != "two-level" else []
        if two_level_ok and args.format == "two-level":
            cands.append(("two-level",) + tile + (8, 4, 3, 3))
    else:
        from voltrix.jit_kernels.spmm import tile_space

        aot = set(capi.tiles(is_f16))
        cands = []
        if args.format != "two-level":
            cands = [("window",) + c for c in sorted({(p["FS"], p["DEPTH"], p["WAVES"], p["SCHED"])
                                                      for p in tile_space(num_feats, in_bytes)
                                                      if (p["FS"], p["DEPTH"], p["WAVES"]) in aot})]
        if two_level_ok:
            fs = 32 if num_feats <= 32 else (64 if num_feats <= 64 else 128)
            # window tile (fs, 3, 4) + panel depth 3 fit one CU together (LDS 103 + 44 KB, registers 136 + 2 x 183)
            plans = [(8, 4, 3, 3 if fs == 128 else 6), (8, 4, 4, 3 if fs == 128 else 6)]
            if local_rows < 100_000 and fs == 128:
                # a rank's shard of a multi-GPU run has too few 512-row panels to fill 256 CUs: shorter panels
                # (measured on a 1/8 shard: profiles/r01/experiment_two_level_shard8.log)
                plans += [(8, 2, 4, 4), (4, 2, 3, 6)]
            for pw, prb, tau, pdepth in plans:
                for sched in (2, 3):
                    cands.append(("two-level", fs, 3, 4, sched, pw, prb, tau, pdepth))
    best = None
    for cand in cands:
        spmm(cand)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            spmm(cand)
        e.record()
        e.synchronize()
        ms = s.elapsed_time(e) / 3
        if world > 1:
            t = torch.tensor([ms], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t)
        if best is None or ms < best[0]:
            best = (ms, cand)
    cand = best[1]
    tile = cand[1:5]

    in_place = args.backend == "nccl"  # only NCCL/RCCL defines the in-place (send == recv + rank * count) form
    overlap = world > 1 and not args.no_overlap
    if world > 1:
        # two copies of the gather buffer: while the SpMM of step k reads one, the all-gather of step k+1 fills the other
        bufs = [gathered, gathered.clone()] if overlap else [gathered]
        sends = [b[rank * rows_padded:(rank + 1) * rows_padded] for b in bufs]
        comm_stream = torch.cuda.Stream(device=device) if overlap else main_stream
        ev_gathered = [None] * len(bufs)   # all-gather into buffer b finished
        ev_consumed = [None] * len(bufs)   # SpMM that read buffer b finished
    step_no = [0]

    def step(record=None):
        """One pass of the hot path: all-gather(B) (N > 1), then the SpMM that consumes exactly that gathered B."""
        if world == 1:
            if record is not None:
                record[0].record()
            spmm(cand)
            if record is not None:
                record[1].record()
            return
        b = step_no[0] % len(bufs)
        step_no[0] += 1
        with torch.cuda.stream(comm_stream):
            if ev_consumed[b] is not None:
                comm_stream.wait_event(ev_consumed[b])
            dist.all_gather_into_tensor(bufs[b], sends[b] if in_place else sends[b].clone())
            ev_gathered[b] = torch.cuda.Event()
            ev_gathered[b].record(comm_stream)
        main_stream.wait_event(ev_gathered[b])
        if record is not None:
            record[0].record()  # HIP events on the launch stream, live inside the timed region
        spmm(cand, bufs[b])
        if record is not None:
            record[1].record()
        ev_consumed[b] = torch.cuda.Event()
        ev_consumed[b].record(main_stream)

    for _ in range(args.warmup):
        step()

    # ---- timed region: exactly K steps between barrier + synchronize, MAX over ranks ------------------------------
    kernel_events = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        step(record=ev)
        kernel_events.append(ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    # ---- outside the timed region: row-sum check of this rank's result against torch ops on the same gathered B -------
    # (validates shard -> column remap -> all-gather -> SpMM end to end on every rank; fp32 index_add reference)
    b_used = gathered if world == 1 else bufs[(step_no[0] - 1) % len(bufs)]
    col_sums = b_used.float().sum(dim=1)
    edge_rows = torch.repeat_interleave(torch.arange(local_rows, device=device),
                                        (local_indptr[1:] - local_indptr[:-1]).long())
    want = torch.zeros(local_rows, dtype=torch.float32, device=device).index_add_(0, edge_rows, col_sums[local_indices.long()])
    got = out.sum(dim=1)
    scale = torch.zeros(local_rows, dtype=torch.float32, device=device).index_add_(
        0, edge_rows, b_used.float().abs().sum(dim=1)[local_indices.long()])
    check_err = float(((got - want).abs() / (scale + 1e-6)).max()) if local_rows else 0.0
    if world > 1:
        t = torch.tensor([check_err], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        check_err = float(t)
    assert check_err < 1e-4, f"row-sum check failed: {check_err}"

    kernel_ms = sum(s.elapsed_time(e) for s, e in kernel_events) / len(kernel_events)
    if world > 1:
        t = torch.tensor([kernel_ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_ms = float(t)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        flop = synth_graphs.flops(nnz, num_feats)
        # roofline of the dominant kernel (spmm_tc16_kernel), per launch on THIS rank's shard:
        # algorithmic bytes = int32 CSR once + B once + C once
        alg_bytes = 4 * (local_nnz + local_rows + 1) + gathered.shape[0] * num_feats * in_bytes + local_rows * num_feats * 4
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        two_level = cand[0] == "two-level"
        if two_level:   # rows actually gathered: 8 per residual TC block + 32 per k-step of the panel plan
            st = two_level_state(*cand[5:8])
            gather_bytes = (8 * st["blocks"] + 32 * st["plan"].num_ksteps) * num_feats * in_bytes
            fmt = {"format": "two-level (voltrix/hybrid.py): window kernel on the residual || panel kernel on the shared "
                             "columns (two streams), + add pass",
                   "panel_rows": st["plan"].panel_rows, "tau": st["plan"].tau, "panel_depth": cand[8],
                   "shared_edge_fraction_rank0": st["plan"].num_shared_edges / max(1, local_nnz),
                   "panel_ksteps_rank0": st["plan"].num_ksteps, "residual_tc_blocks_rank0": st["blocks"],
                   "preprocess_two_level_ms": st["build_ms"]}
        else:
            gather_bytes = 8 * total_blocks * num_feats * in_bytes  # rows gathered from L2 / Infinity Cache / HBM
            fmt = {"format": "window (the reference's block format)"}
        line = {
            "metric": "spmm_gflops",
            "value": flop / (ms_per_step * 1e-3) / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16" if is_f16 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: N={num_nodes} nnz={nnz} (BASELINE.json configs[1] stand-in, "
                            f"SURVEY.md 8d generator, seed {cfg['seed']}) x dense F={num_feats} "
                            f"{'fp16' if is_f16 else 'fp32'} -> fp32",
                "num_nodes": num_nodes, "nnz": nnz, "feat": num_feats, "tc_blocks_rank0": total_blocks,
                "tile": {"fs": tile[0], "depth": tile[1], "waves": tile[2], "balance_schedule_chunk": ORDER_CHUNKS.get(tile[3], 0)},
                "sparse_format": fmt,
                "parallelism": f"row-window shards x{world}" + (
                    " + RCCL all-gather(B) per step" + (" (overlapped with the previous step's SpMM)" if overlap else "")
                    if world > 1 else ""),
                "preprocess_ms": preprocess_ms,
                "rowsum_check_max_rel_err": check_err,
                "hbm_gbs_algorithmic": synth_graphs.algorithmic_bytes(num_nodes, nnz, num_feats, in_bytes)
                / (ms_per_step * 1e-3) / 1e9,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.workload, num_feats, ("f16" if is_f16 else "f32") + ("|two-level" if two_level else ""))
                if (world == 1 and args.scale == 1.0) else None,
                "kernel": ("spmm_tc16_kernel || spmm_panel_kernel, then add_inplace_f32_kernel (HIP events around the three "
                           "launches on the launch stream)") if two_level else "spmm_tc16_kernel",
                "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes,
                "gather_bytes": gather_bytes, "gather_gbs": gather_bytes / (kernel_ms * 1e-3) / 1e9,
                "note": "gather-bound: B rows are served by L2 / Infinity Cache (~8.6-19 TB/s row-gather ceilings), "
                        "see DESIGN.md Roofline",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            vb = vendor_baseline(indptr, indices, num_nodes, num_feats, device)
            if "ms" in vb:
                vb["speedup_of_this_work"] = vb["ms"] / ms_per_step
            line["vendor_gpu_baseline"] = vb
            line["cpu_baseline"] = cpu_baseline(indptr, indices, num_nodes, num_feats)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
