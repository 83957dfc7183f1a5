#!/usr/bin/env python3
"""Headline benchmark: SpMM GFLOP/s + achieved HBM GB/s, reddit-like CSR x dense feat=128 fp16 (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path through the OPERATOR a drop-in caller uses: ``voltrix.spmm(*handle, ...)`` on the
handle ``csr_preprocess`` built (N > 1: preceded by the RCCL all-gather of the dense operand B whose result that SpMM
consumes).  Inputs are resident in HBM before the timed region; preprocessing (CSR -> block format, the two-level
side-car, the unit table, the JIT tile sweep of the first call) happens once, outside it, as in the reference's
protocol (bench/bm_voltrix.py:17,36).  Rank 0 prints ONE JSON line.

Workloads (synth_graphs.py; BASELINE.json's configurations): every N defaults to ``reddit_like`` (configs[1], the
configuration the metric is quoted on) -- the same matrix for every N, row-window shards over the N GPUs + all-gather(B),
i.e. strong scaling of the headline: the per-N values the driver derives its curve from are ONE workload (round 6; the
N > 1 default used to be papers-like, which made N = 1 and N > 1 different graphs).  BASELINE configs[4] -- ``papers_like``,
111 M rows sharded over the N ranks, every rank generating ITS OWN shard -- is measured BESIDE the timed steps of every
N > 1 run and reported as ``config.config5_papers_like`` (all-gather, product and dependent step, MAX over ranks;
``--no-config5`` skips it); as the timed workload: ``--workload papers_like`` (1 GPU: 62 ms/step).
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
    ap.add_argument("--workload", default=None, choices=sorted(synth_graphs.CONFIGS),
                    help="default: reddit_like at every N (round 6: the driver derives its scaling curve from the per-N values, so "
                         "they must be ONE workload -- the headline graph, strong scaling; BASELINE configs[4], papers-like sharded over "
                         "the N ranks, is measured beside it at N > 1: config.config5_papers_like)")
    ap.add_argument("--feat", type=int, default=None, help="feature width (default: the workload's)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the node count (debug only)")
    ap.add_argument("--format", default="auto", choices=["auto", "window", "two-level"],
                    help="auto: what csr_preprocess decides (two-level side-car when enough edges sit in shared columns); "
                         "window: VOLTRIX_HYBRID=0; two-level: VOLTRIX_HYBRID=1 (side-car whenever the plan is not empty)")
    ap.add_argument("--tune", default="default", choices=["default", "full", "none"],
                    help="VOLTRIX_TUNE_SPACE of the first call's tile / schedule sweep")
    ap.add_argument("--no-tuned-defaults", action="store_true",
                    help="measure the first call OFF the shipped buckets: VOLTRIX_TUNED_DEFAULTS=0 and an empty store of "
                         "choices, so the bounded sweep (tuner.py) runs; its cost is first_call_ms / tuner.sweep_seconds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true",
                    help="N > 1: skip the BASELINE configs[4] leg (papers-like sharded over the N ranks, measured after the timed "
                         "steps and reported as config.config5_papers_like)")
    ap.add_argument("--config5-scale", type=float, default=1.0, help="node-count scale of that leg (debug only)")
    ap.add_argument("--no-reference-formats", action="store_true",
                    help="skip the untimed comparison runs (window format alone, cold-cache timing)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for debugging)")
    ap.add_argument("--overlap-steps", action="store_true",
                    help="N > 1: TIME the cross-step overlapped form (the all-gather of step k+1 on a second stream beside the "
                         "SpMM of step k, double-buffered B) -- legitimate only for INDEPENDENT products.  Default: the dependent "
                         "step (all-gather, then the product that consumes it, then the next all-gather: layer l+1's B is layer "
                         "l's C) is ms_per_step and the overlapped figure is reported beside it as config.step_independent_ms")
    ap.add_argument("--no-overlap", action="store_true", help="accepted for compatibility: the dependent step is the default")
    ap.add_argument("--one-device", action="store_true",
                    help="debug: every rank uses cuda:0 (exercises the sharded path on a 1-GPU box, with --backend gloo)")
    ap.add_argument("--gather", default="collective", choices=["auto", "collective", "p2p", "rows"],
                    help="N > 1: the exchange step -- one all_gather_into_tensor (what RCCL picks over xGMI; the DEFAULT), the direct "
                         "schedule written out as world-1 batched point-to-point copies per rank, or only the rows of B the "
                         "shard references (one all-to-all with uneven splits; voltrix/dist.py).  auto: the warm-up "
                         "times collective against p2p (and the referenced-rows operator when the shards reference < 70 %% of "
                         "the remote rows), MAX over ranks, and the timed steps run the fastest -- opt-in since round 6: no "
                         "multi-GPU node has ever run these schedules, and the batched point-to-point candidate has been seen to "
                         "stall between two gloo ranks on one device; the line the driver records must not depend on it")
    ap.add_argument("--rows-below", type=float, default=0.7,
                    help="--gather auto builds and times the referenced-rows operator when the shards reference less than this "
                         "fraction of the remote rows")
    ap.add_argument("--slabs", type=int, default=1,
                    help="N > 1: exchange and multiply B in this many feature slabs (gather of slab j+1 beside the SpMM of "
                         "slab j) instead of overlapping whole steps")
    ap.add_argument("--weighted", action="store_true",
                    help="N = 1: the WEIGHTED product (voltrix.csr_preprocess_weighted / spmm_weighted; SURVEY.md 8f rank 4, no "
                         "reference counterpart) with the symmetric-normalised adjacency of a GCN layer as values, a_ij = "
                         "1 / sqrt(out-degree(i) x in-degree(j)); algorithmic bytes then count 4 more bytes per edge (fp32 values)")
    ap.add_argument("--weighted-plane", action="store_true",
                    help="with --weighted: force the general value plane (separable=False) instead of letting csr_preprocess_weighted "
                         "detect that these values factor as r_i c_j (round 6: then the binary operator runs between two row scalings)")
    ap.add_argument("--backward", action="store_true",
                    help="N = 1: the step is the BACKWARD product dB = A^T dC through the transposed handle (voltrix/autograd.py; "
                         "with --weighted the transposed values): same machinery, same roofline accounting, on A^T")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group, sharded operator, all-gather, barriers) also at world "
                         "size 1: rehearses the RCCL calls of the scaling run on a one-GPU box")
    return ap.parse_args()


def cpu_baseline(indptr, indices, num_rows, num_cols, num_feats, seed=0):
    """torch.sparse.mm (the reference's own oracle call, tests/test_spmm.py:24-29) on the host cores, fp32 -- CPU fp16
    CSR mm is not implemented in torch.  Bounded sample: the whole matrix up to 150 M edges, else a contiguous row
    sample of about 100 M edges."""
    from oracle import torch_ref  # checker / baseline leg only

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    ip, ix = indptr.cpu(), indices.cpu()
    rows, sample = num_rows, "all rows"
    if ix.numel() > 150_000_000:
        rows = int(torch.searchsorted(ip.long(), torch.tensor(100_000_000))) // 16 * 16
        sample = f"contiguous row sample (rows 0..{rows})"
        ip = ip[: rows + 1].clone()
        ix = ix[: int(ip[-1])].clone()
    a = torch_ref.csr_ones(ip, ix, rows, num_cols)
    gen = torch.Generator().manual_seed(seed)
    feat = torch.randn(num_cols, num_feats, generator=gen)
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


def rocsparse_baseline(indptr, indices, num_nodes, feat, ms_per_step):
    """Round 6: rocSPARSE's generic SpMM, every CSR algorithm it offers (buffer-size and preprocess stages outside the timed
    loop, as the reference keeps cuSPARSE's: bench/bm_sparse.py:20-45), on fp32 operands (what the reference's speed-ups are
    quoted on) and on the timed path's own fp16-in / fp32-compute, plus a plain CSR row-gather kernel ("no format").
    harness/bm_rocsparse.cpp: plain HIP + rocSPARSE; warm, back to back, like the timed steps."""
    try:
        from harness import bm_rocsparse

        out = {}
        for label, operand in (("fp32", feat.float()), ("fp16", feat.half())):
            cells = bm_rocsparse.baselines(indptr, indices, num_nodes, operand.contiguous(), iters=5, warmup=2)
            name, ms = bm_rocsparse.best(cells)
            gname, gms = bm_rocsparse.best(cells, prefix="csr_row_gather")
            out[label] = {"ms": {k: (None if v is None else round(v, 4)) for k, v in cells.items()},
                          "rocsparse_best": name, "rocsparse_best_ms": ms,
                          "speedup_of_this_work": None if ms is None else ms / ms_per_step,
                          "csr_row_gather_ms": gms, "speedup_over_csr_row_gather": None if gms is None else gms / ms_per_step}
            del operand
        return out
    except Exception as exc:  # noqa: BLE001  (library missing / out of memory: report, do not fail the bench)
        return {"error": repr(exc)[:200]}


def gather_ceiling(gather_bytes, operand_bytes, kernel_ms):
    """An INDEPENDENT ceiling for the format's gathered bytes: the time the step's gathered rows of B would
    take at the per-CU rates MI355X_MICROARCH.md measures for rows gathered into LDS ('Indexed rows: gather into LDS') -- the
    guide's constants, not this kernel's own rate: 73 GB/s per CU when every row is an L2 hit (the upper end of its 66-73), and
    the rate of where B actually lives when none is (33.5 Infinity Cache up to 128 MB, 30 up to 256 MB, 23.5 HBM).  frac =
    ceiling time / measured kernel time: the share of the step the gathers would need at that rate."""
    if not gather_bytes:
        return None
    beyond = 33.5 if operand_bytes <= (128 << 20) else (30.0 if operand_bytes <= (256 << 20) else 23.5)
    per_cu = gather_bytes / 256
    l2_ms, far_ms = per_cu / 73.0e9 * 1e3, per_cu / (beyond * 1e9) * 1e3
    return {"bytes": gather_bytes, "all_l2": {"gb_s_per_cu": 73.0, "ms": l2_ms, "frac": l2_ms / kernel_ms},
            "none_l2": {"gb_s_per_cu": beyond, "ms": far_ms, "frac": far_ms / kernel_ms},
            "what": "gathered bytes / 256 CUs / the guide's per-CU LDS-gather rate (MI355X_MICROARCH.md 'Indexed rows'); frac = "
                    "ceiling time / measured kernel time (1.0 = the step runs at the guide's gather rate)"}


def gather_model(gather_bytes, l2_hit_frac, operand_bytes, kernel_ms):
    """A MODEL beside the HBM roofline, for the reader: what the CU's row-gather path delivers for this step's hit mix, from the
    per-CU rates MI355X_MICROARCH.md measures for rows gathered into LDS (section 'Indexed rows') -- 66-73 GB/s per CU when the
    rows come out of the XCD's L2, 33.5 / 29-31 / 23-24 when they come from a 38 MB / 151 MB / HBM-sized table -- applied to the
    gathered bytes with the L2 hit fraction of the PMC passes.  The guide calls its rates lower bounds: frac = model time /
    measured time may exceed 1 (the HBM-resident graphs do: 7.3 TB/s against the guide's 6.0-6.1 in-order sweep).  None
    without counters."""
    if l2_hit_frac is None or not gather_bytes:
        return None
    hit_rate = 70.0
    miss_rate = 33.5 if operand_bytes <= (128 << 20) else (30.0 if operand_bytes <= (256 << 20) else 23.5)
    per_cu = gather_bytes / 256
    model_ms = (per_cu * l2_hit_frac / (hit_rate * 1e9) + per_cu * (1.0 - l2_hit_frac) / (miss_rate * 1e9)) * 1e3
    return {"model_ms": model_ms, "frac": model_ms / kernel_ms,
            "rates_gb_s_per_cu": {"l2_hit": hit_rate, "l2_miss": miss_rate},
            "what": "gathered bytes / 256 CUs x (hit fraction / L2-hit gather rate + miss fraction / beyond-L2 gather rate), rates from "
                    "MI355X_MICROARCH.md 'Indexed rows: gather into LDS'; frac = model time / measured kernel time"}


def choose_referenced_rows(op, vdist, local_indptr, local_indices, num_nodes, parts, feat_local, args, device):
    """--gather auto, second half: is the referenced-rows operator (only the rows of B a shard references travel) worth
    building?  Only when the shards reference < 70 % of the remote rows (MAX over ranks).  Then it is built beside the
    all-gather operator and the median of THREE whole steps of each (exchange + product, after two warm-up steps) decides, each
    step MAX over ranks; the ranks agree on every decision through all-reduces.  A rank that fails to build the candidate aborts
    the run (its peers would otherwise wait inside the builder's all-to-all)."""
    world, rank = op.world_size, op.rank
    r0, r1 = parts[rank]
    remote = local_indices[(local_indices < r0) | (local_indices >= r1)]
    referenced = int(torch.unique(remote).numel()) if remote.numel() else 0
    del remote
    frac = torch.tensor([referenced / max(1, num_nodes - (r1 - r0))], dtype=torch.float64, device=device)
    dist.all_reduce(frac, op=dist.ReduceOp.MAX)
    info = {"referenced_fraction_of_remote_rows": float(frac)}
    if float(frac) >= args.rows_below or args.slabs > 1:
        return info
    # from_shard(mode="rows") exchanges its request lists with two all-to-alls: a rank that failed in the middle would leave
    # the others inside a collective it never posts.  No try / except around it -- a failure aborts the run on every rank.
    op_rows = vdist.RowShardedSpMM.from_shard(local_indptr, local_indices, num_nodes, parts, mode="rows",
                                              exchange_at_world_1=args.force_dist)
    num_feats = feat_local.shape[1]

    def whole_step(o, buf):
        o.multiply(o.gather_into(buf, feat_local))

    timings = {}
    for name, o in (("allgather", op), ("rows", op_rows)):
        buf = o._buffer("whole", num_feats, feat_local)
        for _ in range(2):                      # first call: tile choice / unit table of this operator's handle
            whole_step(o, buf)
        samples = []
        for _ in range(3):                      # median of three whole steps, each MAX over ranks
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            whole_step(o, buf)
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) * 1e3], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            samples.append(float(t))
        timings[name] = sorted(samples)[1]
    info["whole_step_ms"] = timings
    if timings["rows"] < timings["allgather"]:
        info["picked"] = "rows"
        info["op_rows"] = op_rows
    return info


def measured_counters(key, sources_hash):
    """PMC-derived figures of the step (profiles/traffic.json, written from rocprofv3 passes of THIS command): fabric-side
    bytes per step and the matrix-core busy fraction -- only for the exact configuration `key` names (workload, width,
    dtype, format, tile, schedule) AND only while the entry was measured on the kernel sources of this tree (its
    ``sources_hash`` = voltrix.jit.compiler.get_kernel_sources_version(), written by harness/pmc_summarize.py): the counters
    need their own profiling passes and cannot be read live, so an entry that predates a kernel edit is refused, not replayed.
    -> (entry or None, why not)."""
    try:
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            entry = json.load(f).get("runs", {}).get(key)
    except (OSError, ValueError, KeyError):
        entry = None
    if entry is None:
        return None, f"none for this exact configuration ({key}): PMC passes are separate runs (profiles/)"
    if entry.get("sources_hash") != sources_hash:
        return None, (f"stale: the entry for {key} ({entry.get('source')}) was measured on kernel sources "
                      f"{entry.get('sources_hash')}, this tree is {sources_hash} -- re-run harness/pmc_bench.sh")
    return entry, None


def config5_leg(world, rank, device, scale=1.0, steps=3):
    """BASELINE configs[4] beside the timed headline steps of an N > 1 run: the papers-like stand-in (111 M nodes, 1.6 G edges, F = 128
    fp16) row-sharded over the N ranks -- every rank generates and preprocesses its own shard on its device, one RCCL all-gather of B
    per step, then the local product (voltrix.dist.RowShardedSpMM, the collective schedule).  Returns the figures SURVEY 8(d) asks for
    (all-gather and product separately and combined, MAX over ranks); any failure is reported as a string instead of ending the run."""
    import voltrix  # noqa: F401
    from voltrix import dist as vdist

    workload, num_feats = "papers_like", 128
    try:
        deg = synth_graphs.target_degrees(workload, device=device, scale=scale)
        num_nodes, nnz = deg.numel(), int(deg.sum())
        full_indptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=device)
        full_indptr[1:] = torch.cumsum(deg, 0)
        parts = vdist.partition_rows(full_indptr, num_nodes, world)
        del deg, full_indptr
        r0, r1 = parts[rank]
        t0 = time.perf_counter()
        local_indptr, local_indices, _ = synth_graphs.generate(workload, device=device, scale=scale, rows=(r0, r1))
        torch.cuda.synchronize()
        generate_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        op = vdist.RowShardedSpMM.from_shard(local_indptr, local_indices, num_nodes, parts, mode="collective")
        torch.cuda.synchronize()
        preprocess_ms = (time.perf_counter() - t0) * 1e3
        op.handle[1].hash_tag = f"bench/{workload}/config5/s{scale}/r{rank}of{world}"
        gen = torch.Generator(device=device).manual_seed(4321 + rank)
        feat_local = torch.randn(r1 - r0, num_feats, generator=gen, device=device, dtype=torch.float32).half()
        buf = op._buffer("whole", num_feats, feat_local)

        def timed(fn, reps):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)

        out = [None]

        def step():
            out[0] = op.multiply(op.gather_into(buf, feat_local))

        step()          # first call: tile choice, tables, RCCL channels
        step()
        allgather_ms = timed(lambda: op.gather_into(buf, feat_local), steps)
        spmm_ms = timed(lambda: op.multiply(buf), steps)
        step_ms = timed(step, steps)
        # row sums: C 1 = A (B 1) on this rank's rows, against the degrees times the column means it implies (cheap sanity check)
        ones = torch.ones(r1 - r0, 8, device=device, dtype=torch.float16)
        deg_local = (local_indptr[1:] - local_indptr[:-1]).float()
        got = op.multiply(op.gather_into(op._buffer("ones", 8, ones), ones))[:, 0]
        rel = float(((got - deg_local).abs().max() / deg_local.max().clamp(min=1)))
        return {"workload": f"{workload}: N={num_nodes} nnz={nnz} x F={num_feats} fp16, row-window shards x{world} (BASELINE.json configs[4])",
                "allgather_ms": allgather_ms, "local_spmm_ms": spmm_ms, "step_ms": step_ms,
                "gflops": synth_graphs.flops(nnz, num_feats) / (step_ms * 1e-3) / 1e9,
                "allgather_bytes_received_per_rank": op.exchange_bytes_received(num_feats, 2),
                "shard_rows_rank0": r1 - r0, "shard_nnz_rank0": int(local_indices.numel()), "generate_s_rank0": generate_s,
                "preprocess_ms_rank0": preprocess_ms, "degree_check_max_rel_err_rank0": rel, "steps": steps,
                "what": "dependent step = all-gather(B) then the product that consumes it, MAX over ranks; measured after the timed "
                        "headline steps, not part of `value`"}
    except Exception as exc:  # noqa: BLE001 -- a side measurement must never take the bench line down
        return {"error": f"{type(exc).__name__}: {str(exc)[:400]}"}


def spawn_ranks(args):
    """``python bench.py --gpus N`` (N > 1) outside torch.distributed.run: start the N ranks as FRESH child processes -- one
    ``python -m torch.distributed.run`` with this command line -- before this process has touched the GPU, relay their output
    (rank 0 prints the JSON line) and exit with their code.  Nothing is re-executed in place: a process that initialised the GPU
    must never exec."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env, stdin=subprocess.DEVNULL)


def main():
    args = parse_args()
    if os.getenv("VOLTRIX_BENCH_STACKS_AFTER"):      # debugging aid: every thread's Python stack on stderr after that many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["VOLTRIX_BENCH_STACKS_AFTER"]), repeat=False, file=sys.stderr)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1 or args.force_dist
    # Only the JSON line may reach stdout.  The communication libraries print banners there (RCCL: version / HIP / hostname, five
    # lines per rank; gloo: "Rank r is connected to ..."): in a distributed run everything written to fd 1 goes to stderr and the
    # line is written to the saved descriptor at the end.
    result_fd = None
    if distributed:
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
    if distributed:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    os.environ["VOLTRIX_TUNE_SPACE"] = args.tune
    if args.no_tuned_defaults:
        import tempfile
        os.environ["VOLTRIX_TUNED_DEFAULTS"] = "0"
        os.environ["VOLTRIX_TUNED_STORE"] = os.path.join(tempfile.mkdtemp(prefix="voltrix_bench_"), f"tuned_rank{rank}.json")
    if args.format != "auto":
        os.environ["VOLTRIX_HYBRID"] = "0" if args.format == "window" else "1"
        os.environ.setdefault("VOLTRIX_HYBRID_MIN_SHARE", "0")

    import voltrix
    from voltrix import dist as vdist
    from voltrix.jit_kernels import jit_tuner
    from voltrix.jit_kernels.spmm import ORDER_CHUNKS, SCHED_PAIRS, SCHED_STREAM, SCHED_UNITS, slab_launches

    workload = args.workload or "reddit_like"
    config_index = {"cora_like": 0, "reddit_like": 1, "reddit_uniform": 1, "reddit_shuffled": 1, "reddit_sbm": 1,
                    "reddit_sbm_shuffled": 1, "products_like": 2,
                    "products_shuffled": 2, "powerlaw_4m": 3, "papers_like": 4}.get(workload)
    cfg = synth_graphs._resolve(workload)    # label-shuffled variants inherit their base config
    num_feats = args.feat or cfg["feat"]
    is_f16 = args.dtype == "f16"
    in_bytes = 2 if is_f16 else 4

    # ---- synthetic graph: every rank generates ITS OWN row-window shard (same degree sequence on every rank) ---------
    deg = synth_graphs.target_degrees(workload, device=device, scale=args.scale)
    num_nodes = deg.numel()
    nnz = int(deg.sum())
    full_indptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=device)
    full_indptr[1:] = torch.cumsum(deg, 0)
    parts = vdist.partition_rows(full_indptr, num_nodes, world)
    del deg, full_indptr
    r0, r1 = parts[rank]
    rows_padded = max(1, max(p[1] - p[0] for p in parts))
    local_indptr, local_indices, _ = synth_graphs.generate(workload, device=device, scale=args.scale,
                                                           rows=(r0, r1) if distributed else None)
    local_rows, local_nnz = r1 - r0, local_indices.numel()
    num_cols = world * rows_padded if world > 1 else num_nodes   # ids index the gathered B (padded shards) when sharded

    # ---- preprocess: the operator a drop-in caller uses.  N > 1 (and --force-dist): voltrix.dist.RowShardedSpMM built from
    # ---- this rank's OWN device-resident shard (no full graph anywhere, no host round trip) -- the advertised class IS the
    # ---- measured one
    op = None
    whandle = edge_values = None
    if args.weighted:
        assert not distributed, "--weighted is a single-GPU run"
        d_out = (local_indptr[1:] - local_indptr[:-1]).float().clamp(min=1)
        d_in = torch.bincount(local_indices.long(), minlength=num_cols).float().clamp(min=1)
        edge_values = torch.repeat_interleave(d_out.rsqrt(), (local_indptr[1:] - local_indptr[:-1]).long()) * d_in.rsqrt()[local_indices.long()]
        # graphs too large for the automatic detection of v_ij = r_i c_j (weighted.SEPARABLE_MAX_EDGES): the caller states the factors,
        # as a GCN layer that builds the normalisation itself would
        stated_scales = (d_out.rsqrt(), d_in.rsqrt()) if local_nnz > (1 << 29) and not args.weighted_plane else None
        del d_out, d_in
    if args.backward:   # the transposed operator: what autograd.SpMM builds for the gradient with respect to B
        assert not distributed, "--backward is a single-GPU run"
        from voltrix import capi as _capi
        from voltrix.weighted import transpose_weighted

        if edge_values is not None:
            local_indptr, local_indices, edge_values = transpose_weighted(local_indptr, local_indices, edge_values, local_rows, num_cols)
        else:
            local_indptr, local_indices = _capi.csr_transpose(local_indptr.contiguous(), local_indices.contiguous(), local_rows, num_cols)
        torch.cuda.empty_cache()
    preprocess_ms = None
    for _ in range(2):   # first call pays library load / allocator warm-up; report the second (host wall clock, sync'd)
        if args.weighted:   # the first handle (62 GB with its value plane at the papers-like size) must not outlive its purpose
            whandle = handle = None
            torch.cuda.empty_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if distributed:
            op = vdist.RowShardedSpMM.from_shard(local_indptr, local_indices, num_nodes, parts,
                                                 mode="collective" if args.gather == "auto" else args.gather,
                                                 slabs=args.slabs, exchange_at_world_1=args.force_dist)
            handle = op.handle
        elif args.weighted:
            if stated_scales is not None and not args.backward:
                whandle = voltrix.csr_preprocess_weighted(local_indptr, local_indices, None, local_rows, num_cols=num_cols,
                                                          row_scale=stated_scales[0], col_scale=stated_scales[1])
            else:
                whandle = voltrix.csr_preprocess_weighted(local_indptr, local_indices, edge_values, local_rows, num_cols=num_cols,
                                                          separable=False if args.weighted_plane else "auto")
            handle = (whandle.blk_offsets, whandle.hspa_packed, whandle.hind)
        else:
            handle = voltrix.csr_preprocess_device(local_indptr, local_indices, local_rows, num_cols=num_cols)
        torch.cuda.synchronize()
        preprocess_ms = (time.perf_counter() - t0) * 1e3
    if args.weighted:
        torch.cuda.empty_cache()   # the plane builder's chunk temporaries
    handle[1].hash_tag = f"bench/{workload}/s{args.scale}/r{rank}of{world}" + ("/transposed" if args.backward else "")
    # the kernels read a value plane (else: binary kernels between two row scalings; a separable handle still takes the plane where the
    # scalings would move more bytes than it: weighted.separable_pays)
    from voltrix.weighted import separable_pays
    weighted_plane = args.weighted and not (whandle.separable and separable_pays(whandle, num_feats, in_bytes))
    two = voltrix.two_level_of(handle[1])
    total_blocks = int(handle[0][-1])
    from voltrix import hybrid as vhybrid

    handle_bytes = {"reference_handle": vhybrid.handle_bytes(handle),
                    "two_level_side_car": vhybrid.two_level_bytes(two) if two is not None else 0}

    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    feat_local = torch.randn(local_rows, num_feats, generator=gen, device=device,
                             dtype=torch.float32).to(torch.float16 if is_f16 else torch.float32)
    # ---- --gather auto: the exchange schedule is chosen from measurement, identically on every rank -------------------
    gather_mode = args.gather
    exchange_choice = None
    if distributed and args.gather == "auto":
        exchange_choice = {"candidates_ms": op.choose_exchange(feat_local)}      # collective vs p2p: same buffer layout
        gather_mode = op.mode
        if world > 1:
            exchange_choice.update(choose_referenced_rows(op, vdist, local_indptr, local_indices, num_nodes, parts, feat_local,
                                                          args, device))
            if exchange_choice.get("picked") == "rows":
                op = exchange_choice.pop("op_rows")
                handle = op.handle
                handle[1].hash_tag = f"bench/{workload}/s{args.scale}/r{rank}of{world}/rows"
                two = voltrix.two_level_of(handle[1])
                total_blocks = int(handle[0][-1])
                gather_mode = "rows"
            exchange_choice.pop("op_rows", None)
        exchange_choice["picked"] = gather_mode
    if world > 1 and gather_mode == "rows":
        gathered = torch.zeros(op.compact_rows, num_feats, dtype=feat_local.dtype, device=device)   # own rows | referenced rows
        gathered[:local_rows].copy_(feat_local)
    elif world > 1:
        gathered = torch.zeros(world * rows_padded, num_feats, dtype=feat_local.dtype, device=device)
        gathered[rank * rows_padded:rank * rows_padded + local_rows].copy_(feat_local)
    else:
        gathered = feat_local
    main_stream = torch.cuda.current_stream()
    out_holder = [None]

    def spmm(b):
        """The operator call of a drop-in caller (allocates its output, like the reference's spmm.py:101)."""
        if op is not None:
            out_holder[0] = op.multiply(b)      # = voltrix.spmm(*op.handle, ...) on the gathered B
        elif whandle is not None:
            out_holder[0] = voltrix.spmm_weighted(whandle, b)
        else:
            out_holder[0] = voltrix.spmm(*handle, num_nodes=local_rows, num_edges=local_nnz, feat=b)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spmm(gathered)   # first call: tile / schedule from the persisted choice, the bucket, or a sweep; unit table; side stream
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t0) * 1e3
    tuner_stats = dict(jit_tuner.stats)

    slab_pipeline = distributed and args.slabs > 1
    can_overlap = distributed and not slab_pipeline
    overlap_timed = can_overlap and args.overlap_steps       # default: the DEPENDENT step is the one that is timed
    if distributed:
        # two copies of the gather buffer: while the SpMM of step k reads one, the all-gather of step k+1 fills the other
        # (the independent-products form; the dependent step uses one buffer and one stream)
        bufs = [gathered, gathered.clone()] if can_overlap else [gathered]
        comm_stream = torch.cuda.Stream(device=device) if can_overlap else main_stream
        ev_gathered = [None] * len(bufs)   # all-gather into buffer b finished
        ev_consumed = [None] * len(bufs)   # SpMM that read buffer b finished
    step_no = [0]

    def step(record=None, overlap=None):
        """One pass of the hot path: all-gather(B) (N > 1), then the SpMM that consumes exactly that gathered B.
        ``overlap`` False (the timed default): everything on the caller's stream, step k + 1 starts when step k's product is done
        -- a chain of layers.  True: the all-gather of step k + 1 beside the product of step k (independent products)."""
        overlap = overlap_timed if overlap is None else overlap
        if not distributed:
            if record is not None:
                record[0].record()
            spmm(gathered)
            if record is not None:
                record[1].record()
            return
        if slab_pipeline:       # the operator's own pipeline: slab j + 1 travels while slab j is multiplied
            if record is not None:
                record[0].record()
            out_holder[0] = op(feat_local)
            if record is not None:
                record[1].record()
            return
        if not overlap:
            op.gather_into(bufs[0], feat_local)
            if record is not None:
                record[0].record()
            spmm(bufs[0])
            if record is not None:
                record[1].record()
            step_no[0] = 1
            return
        b = step_no[0] % len(bufs)
        step_no[0] += 1
        with torch.cuda.stream(comm_stream):
            if ev_consumed[b] is not None:
                comm_stream.wait_event(ev_consumed[b])
            op.gather_into(bufs[b], feat_local)
            ev_gathered[b] = torch.cuda.Event()
            ev_gathered[b].record(comm_stream)
        main_stream.wait_event(ev_gathered[b])
        if record is not None:
            record[0].record()  # HIP events on the launch stream, live inside the timed region
        spmm(bufs[b])
        if record is not None:
            record[1].record()
        ev_consumed[b] = torch.cuda.Event()
        ev_consumed[b].record(main_stream)

    for _ in range(args.warmup):
        step()

    # ---- timed region: exactly K steps between barrier + synchronize, MAX over ranks ------------------------------
    kernel_events = []
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        step(record=ev)
        kernel_events.append(ev)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    out = out_holder[0]

    # ---- outside the timed region: row-sum check of this rank's result against torch ops on the same gathered B -------
    # (validates shard -> column remap -> all-gather -> SpMM end to end on every rank; fp32 index_add reference, chunked)
    if not distributed:
        b_used = gathered
    elif slab_pipeline:
        b_used = op.gather_into(bufs[0], feat_local)
    else:
        b_used = bufs[(step_no[0] - 1) % len(bufs)] if overlap_timed else bufs[0]
    col_sums = torch.empty(b_used.shape[0], dtype=torch.float32, device=device)
    col_abs = torch.empty(b_used.shape[0], dtype=torch.float32, device=device)
    for q in range(0, b_used.shape[0], 1 << 22):      # by row chunks: a float copy of B is 53 GB at the papers-like size
        x = b_used[q:q + (1 << 22)].float()
        col_sums[q:q + (1 << 22)] = x.sum(dim=1)
        col_abs[q:q + (1 << 22)] = x.abs().sum(dim=1)
        del x
    want = torch.zeros(local_rows, dtype=torch.float32, device=device)
    scale = torch.zeros(local_rows, dtype=torch.float32, device=device)
    row_of_edge_chunk = 1 << 27
    ip64 = local_indptr.long()
    # the shard's column ids are global; the gathered B is laid out in padded shards (voltrix.dist.remap_columns)
    if world > 1 and gather_mode == "rows":   # the compact buffer: positions the operator computed at setup
        check_indices = op.compact_ids(local_indices)
    else:
        check_indices = vdist.remap_columns(local_indices, parts, rows_padded) if world > 1 else local_indices
    for e0 in range(0, local_nnz, row_of_edge_chunk):
        e1 = min(local_nnz, e0 + row_of_edge_chunk)
        cols = check_indices[e0:e1].long()
        rows = torch.searchsorted(ip64, torch.arange(e0, e1, device=device), right=True) - 1
        w_e = edge_values[e0:e1] if edge_values is not None else 1.0
        want.index_add_(0, rows, col_sums[cols] * w_e)
        scale.index_add_(0, rows, col_abs[cols] * w_e)
        del cols, rows
    got = torch.zeros(local_rows, dtype=torch.float32, device=device)
    for q in range(0, local_rows, 1 << 22):
        got[q:q + (1 << 22)] = out[q:q + (1 << 22)].sum(dim=1)
    check_err = float(((got - want).abs() / (scale + 1e-6)).max()) if local_rows else 0.0
    if distributed:
        t = torch.tensor([check_err], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        check_err = float(t)
    # edge values enter the MFMA as fp16: each term carries their rounding (2^-11 relative to |w b|), so the weighted
    # bound is that unit roundoff on top of the unweighted bound
    check_tol = 1e-4 + (0.0 if edge_values is None else 2.0 ** -11)
    assert check_err < check_tol, f"row-sum check failed: {check_err} (tolerance {check_tol})"

    kernel_ms = sum(s.elapsed_time(e) for s, e in kernel_events) / len(kernel_events)
    if distributed:
        t = torch.tensor([kernel_ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_ms = float(t)

    # ---- per-kernel durations of a step (outside the timed region): every launch of the operator bracketed by its own
    # ---- HIP event pair on its own stream (voltrix.utils.KernelTimer); the panel kernel runs beside the window kernel
    from voltrix.utils import KernelTimer

    with KernelTimer() as timer:
        for _ in range(10):
            step()
    kernels_ms = {k: round(v[1], 4) for k, v in timer.summary().items()}

    # ---- the like-for-like step for the fp32 baselines: the same operator call on fp32 features, outside
    # ---- the timed region (the reference feeds fp32; here they become scaled fp16 or exact fp32 tiles: `config.fp32_in_runs_as`)
    fp32_in_ms = None
    if world == 1 and is_f16 and whandle is None and not args.no_cpu_baseline and gathered.numel() * 4 < (8 << 30):
        feat32 = gathered.float()
        call32 = lambda: voltrix.spmm(*handle, num_nodes=local_rows, num_edges=local_nnz, feat=feat32)  # noqa: E731
        for _ in range(3):
            call32()
        ev32 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev32[0].record()
        for _ in range(10):
            call32()
        ev32[1].record()
        ev32[1].synchronize()
        fp32_in_ms = ev32[0].elapsed_time(ev32[1]) / 10
        del feat32

    # ---- what the operator ran (the tuner's choice), for the record -----------------------------------------------------
    from voltrix.spmm.spmm import fp32_mode

    fp32_as = fp32_mode(handle[1], local_rows, num_feats)   # what fp32 features become on this handle: "fp16" (scaled cast) | "exact"

    def tuned(hspa_packed, beside_panel):
        from voltrix.jit_kernels.spmm import feature_hash

        keys = {"feature_hash": feature_hash(hspa_packed), "embedding_dim": num_feats,
                "dtype": str(torch.float16 if is_f16 else torch.float32),
                "device": torch.cuda.get_device_name(device), "two_level": bool(beside_panel), "weighted": bool(weighted_plane)}
        if not is_f16 and fp32_as == "fp16":
            keys["dtype"] = str(torch.float16)   # fp32 features run as scaled fp16
        return jit_tuner.tuned_point("spmm_kernel", keys)

    padded_width = (num_feats + 7) // 8 * 8
    operand_dtype = str(torch.float16) if (is_f16 or fp32_as == "fp16") else str(torch.float32)
    hybrid_env = os.getenv("VOLTRIX_HYBRID", "auto")
    if two is None or hybrid_env in ("0", "off"):
        used_two = False
    elif hybrid_env == "tune":   # opt-in: the operator timed both forms on its first call and kept the faster
        used_two = two.format_choice.get((padded_width, operand_dtype)) == "two-level"
    else:   # auto (csr_preprocess decided from the plan's statistics: the side-car exists only when it is to be used) / forced
        used_two = True
    point = tuned(two.hspa_packed if used_two else handle[1], used_two)

    def sched_name(p):
        s = p.get("SCHED")
        if s == SCHED_UNITS:
            return "unit table (windows cut at 1.5 x the median length, longest first) + combine pass"
        if s == SCHED_PAIRS:
            return ("unit table (windows cut at 1.25 x the median length, longest first), two units per wave "
                    "(spmm_tc16_pair_kernel) + combine pass")
        if s == SCHED_STREAM:
            return ("stream of stages (spmm_stream_kernel): a wave walks a run of consecutive windows through one ring, 16-byte "
                    "stores from the loop; windows above the cut length in interleaved units + combine pass")
        return "natural window order" if s == 0 else f"balance schedule, chunk {ORDER_CHUNKS.get(s)}"

    # ---- untimed comparison runs (N = 1): the window format alone, and a cold-cache timing --------------------------------
    extras = {"first_call_ms": first_call_ms, "handle_bytes": handle_bytes,
              "tuner": dict(tuner_stats, note="sweeps / candidates timed / choices taken from the persisted exact key / from the "
                                              "persisted graph-statistics bucket, up to and including the first call")}
    if distributed:
        # the two halves of a step on their own ("report all-gather time and SpMM time separately + combined"):
        # three all-gathers back to back on the communication stream, MAX over ranks; the local SpMM is roofline.kernel_ms
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(main_stream):
            for _ in range(3):
                op.gather_into(bufs[0], feat_local)
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        extras["allgather_ms"] = float(t)
        extras["allgather_mode"] = gather_mode + (f", {args.slabs} feature slabs pipelined" if slab_pipeline else "")
        if exchange_choice is not None:
            extras["exchange_choice"] = exchange_choice   # what the warm-up measured (MAX over ranks) and what it kept
        extras["allgather_bytes_received_per_rank"] = op.exchange_bytes_received(num_feats, gathered.element_size())
        extras["local_spmm_ms"] = kernel_ms
        if can_overlap:
            # the other form of the step, K steps between barriers like the timed region (MAX over ranks): with the dependent
            # step timed (default) this is what INDEPENDENT products would get from overlapping the exchange of step k + 1
            # with the product of step k; with --overlap-steps it is the dependent chain
            other = not overlap_timed
            torch.cuda.synchronize()    # the two forms use the buffers from different streams: a clean hand-over
            for _ in range(2):
                step(overlap=other)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(overlap=other)
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.tensor([(time.perf_counter() - t0) / args.steps * 1e3], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()
            extras["step_independent_ms" if other else "step_dependent_ms"] = float(t)
            extras["timed_step"] = ("dependent: all-gather(B), then the product that consumes it, on one stream; the next step "
                                    "starts when this one's product is done" if not overlap_timed else
                                    "independent products: the all-gather of step k + 1 beside the product of step k")
        # what the step should take on a fully connected xGMI node (profiles/HISTORY.md section 6), to read the measured one against
        extras["predicted_ms"] = {k: round(v, 3) for k, v in vdist.predicted_step_ms(
            world, rows_padded * num_feats * gathered.element_size(), kernel_ms).items()}
        # the 1-GPU point of THIS workload's strong-scaling curve (the driver's own N = 1 run is the headline workload,
        # reddit-like): measured on one MI355X with the same command and --gpus 1 --workload <this one>, kept in profiles/
        ref_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03", f"bench_{workload}_f{num_feats}.json")
        if args.scale == 1.0 and is_f16 and os.path.exists(ref_path):
            ref = [json.loads(ln) for ln in open(ref_path) if ln.startswith("{")][-1]
            extras["single_gpu_same_workload"] = {"ms_per_step": ref["ms_per_step"], "value": ref["value"], "unit": ref["unit"],
                                                  "source": os.path.relpath(ref_path, os.path.dirname(os.path.abspath(__file__)))}
    if world == 1 and not args.no_reference_formats:
        def time_ms(fn, iters=10):
            fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            e.synchronize()
            return s.elapsed_time(e) / iters

        if two is not None:
            os.environ["VOLTRIX_HYBRID"] = "0"
            wh = tuple(t for t in handle)   # same tensors; the side-car is ignored with VOLTRIX_HYBRID=0
            extras["window_format_ms"] = time_ms(lambda: voltrix.spmm(*wh, num_nodes=local_rows, num_edges=local_nnz,
                                                                      feat=gathered))
            extras["window_format_choice"] = dict(tuned(handle[1], False))
            os.environ["VOLTRIX_HYBRID"] = hybrid_env
            extras["format_choice"] = ({f"F={k[0]} {k[1]}": v for k, v in two.format_choice.items()} if hybrid_env == "tune"
                                       else f"two-level: decided by csr_preprocess from the plan's count phase "
                                            f"({two.plan.num_shared_edges / max(1, local_nnz):.3f} of the edges in shared columns, "
                                            f"threshold {vhybrid.min_shared_fraction()})")
            if (used_two and is_f16 and two.plan.waves == vhybrid.DEFAULT_WAVES
                    and two.plan.row_blocks == vhybrid.DEFAULT_ROW_BLOCKS):
                # for the record: the same product as ONE launch (spmm_fused_kernel; not the default form, profiles/HISTORY.md 3.7)
                if two.fused is None:
                    two.fused = vhybrid.build_fused_records(two.blk_offsets, two.hspa_packed, two.hind, two.num_nodes)
                fused_out = torch.empty(local_rows, num_feats, dtype=torch.float32, device=device)
                extras["one_launch_form_ms"] = time_ms(lambda: vhybrid.launch_fused(two.plan, two.fused, gathered, fused_out))
                extras["one_launch_form_max_abs_diff"] = float((fused_out - out).abs().max())
                two.fused = None
                del fused_out
        # the reference's boundary hands csr_preprocess HOST buffers (CPU int32 indptr / indices, reference spmm/spmm.py:21-22): the same
        # handle built from them, H2D copy over PCIe included (pageable memory, as a caller's numpy / torch CPU arrays are) -- never part of
        # `value`, whose operands are resident in HBM
        if not args.weighted and not args.backward and local_nnz <= (1 << 28):
            h_indptr, h_indices = local_indptr.cpu(), local_indices.cpu()
            host_ms = []
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                from_host = voltrix.csr_preprocess(h_indptr, h_indices, local_rows)
                torch.cuda.synchronize()
                host_ms.append((time.perf_counter() - t0) * 1e3)
            same = all(torch.equal(a, b) for a, b in zip(from_host, handle)) if num_cols == local_rows else None
            extras["preprocess_from_host"] = {"ms": host_ms[-1], "h2d_bytes": 4 * (local_nnz + local_rows + 1), "same_handle": same,
                                              "what": "voltrix.csr_preprocess(indptr.cpu(), indices.cpu(), N): the reference's call with "
                                                      "host buffers, H2D copy included; second of two calls"}
            del h_indptr, h_indices, from_host
        # cold caches: 512 MB written between steps (more than L2 + the 256 MB Infinity Cache), as the reference's
        # bench_kineto does with 256 MB for a 50 MB L2 (utils.py:277-281); the headline number is the warm-cache one
        flush = torch.empty(512 << 20, dtype=torch.uint8, device=device)
        cold = []
        for _ in range(5):
            flush.zero_()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            spmm(gathered)
            e.record()
            e.synchronize()
            cold.append(s.elapsed_time(e))
        extras["cold_cache_ms"] = sorted(cold)[len(cold) // 2]
        del flush

    if world > 1 and not args.no_config5 and workload == "reddit_like":
        extras["config5_papers_like"] = config5_leg(world, rank, device, scale=args.config5_scale)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        flop = synth_graphs.flops(nnz, num_feats)
        # roofline of the step's kernels, per step on THIS rank's shard:
        # algorithmic bytes = int32 CSR once + B once + C once
        alg_bytes = ((8 if args.weighted else 4) * local_nnz + 4 * (local_rows + 1) + gathered.shape[0] * num_feats * in_bytes
                     + local_rows * num_feats * 4)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        if used_two:   # rows actually gathered: 8 per residual TC block + 32 per k-step of the panel plan
            resid_blocks = int(two.blk_offsets[-1])
            gather_bytes = (8 * resid_blocks + 32 * two.plan.num_ksteps) * num_feats * in_bytes
            fmt = {"format": "two-level (voltrix/hybrid.py): panel kernel on the shared columns || window kernel on the "
                             "residual (two streams), float-atomic epilogues onto a zero-filled C (no second buffer, no "
                             "add pass); cut windows of the unit table summed by the combine pass after the join",
                   "join": "atomic",
                   "panel_rows": two.plan.panel_rows, "tau": two.plan.tau,
                   "shared_edge_fraction_rank0": two.plan.num_shared_edges / max(1, local_nnz),
                   "panel_ksteps_rank0": two.plan.num_ksteps, "residual_tc_blocks_rank0": resid_blocks,
                   # round 4: the schedules of the step (data, not timing): XCD ranges of equal work, long panels in pieces
                   "xcd_ranges": "equal work" if two.plan.xcd_ptr is not None else "equal counts",
                   "panel_pieces": None if two.plan.parts is None else {
                       "bound_ksteps": two.plan.parts.cap, "pieces": two.plan.parts.num_parts,
                       "cut_panels": two.plan.parts.num_cuts, "partial_tiles": two.plan.parts.num_slots}}
            kernels = ("memset(C) ; spmm_panel_kernel || "
                       + ("spmm_tc16_pair_kernel" if point.get("SCHED") == SCHED_PAIRS else "spmm_tc16_kernel")
                       + " ; combine_partials_kernel"
                       + (" ; combine_panel_partials_kernel" if two.plan.parts is not None and two.plan.parts.num_slots else ""))
        else:
            gather_bytes = 8 * total_blocks * num_feats * in_bytes  # rows gathered from L2 / Infinity Cache / HBM
            fmt = {"format": "window (the reference's block format)" + (
                " + value plane [T, 16, 8] in the operand's 16-bit type (voltrix/weighted.py: 256 B per TC block, fetched by one "
                "more LDS-DMA per stage; values = symmetric-normalised adjacency)" if weighted_plane else "")}
            kernels = ("spmm_tc16_pair_kernel" if point.get("SCHED") == SCHED_PAIRS else
                       ("spmm_stream_kernel" if point.get("SCHED") == SCHED_STREAM else "spmm_tc16_kernel")) + (
                " ; combine_partials_kernel" if point.get("SCHED") in (SCHED_UNITS, SCHED_PAIRS, SCHED_STREAM) else "")
        if args.weighted and not weighted_plane:
            fmt["values"] = ("separable: v_ij = r_i c_j " + ("stated by the caller (row_scale / col_scale: above weighted.SEPARABLE_MAX_EDGES "
                                                              "the detection is not tried)" if stated_scales is not None and not args.backward
                                                              else "detected by csr_preprocess_weighted (exact edge-by-edge check)") + "; the step is "
                             "scale_rows(B, c) ; the binary operator ; scale_rows(C, r) -- no value plane (voltrix/weighted.py)")
            kernels = "scale_rows_kernel(B) ; " + kernels + " ; scale_rows_kernel(C)"
        # round 6: handles of short windows may run the CSR row-gather kernel instead (voltrix.spmm measured both on its first call)
        from voltrix import sidecar as vsidecar

        csr_sc = vsidecar.lookup_csr(handle[1]) if not args.weighted else None
        used_csr = csr_sc is not None and csr_sc.choice.get((int(gathered.shape[1]), str(gathered.dtype))) == "csr"
        if used_csr:
            gather_bytes = local_nnz * num_feats * in_bytes
            fmt = {"format": "CSR row-gather kernel (spmm_csr_kernels.hpp: no block format, no matrix cores; chosen by measurement "
                             "against the block-format path on the first call)"}
            kernels = "spmm_csr_rows_kernel"
        tile_desc = {"fs": point.get("FS"), "depth": point.get("DEPTH"), "waves": point.get("WAVES"),
                     "schedule": sched_name(point), "sched": point.get("SCHED"),
                     # wide operands: one launch per 256-byte group of column slabs (spmm_kernels.hpp::slab_launch_group)
                     "launches_per_step": slab_launches(num_feats, point.get("FS") or 128, in_bytes, num_nodes)}
        counter_key = (f"{workload}{'+values' if weighted_plane else ('+scales' if args.weighted else '')}{'^T' if args.backward else ''}|F{num_feats}|{args.dtype}|{'csr' if used_csr else ('two-level' if used_two else 'window')}|"
                       f"{point.get('FS')},{point.get('DEPTH')},{point.get('WAVES')}|sched{point.get('SCHED')}")
        from voltrix.jit.compiler import get_kernel_sources_version

        sources_hash = get_kernel_sources_version()
        counters, why_not = (measured_counters(counter_key, sources_hash) if (world == 1 and args.scale == 1.0)
                             else (None, "counters are kept for the full-size single-GPU configurations only"))
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
                "workload": f"{workload}: N={num_nodes} nnz={nnz} ("
                            + (f"BASELINE.json configs[{config_index}] stand-in, SURVEY.md 8d generator" if config_index is not None
                               else "stand-in for a graph of the reference's evaluation set, bench/plot.py:8; synth_graphs.py")
                            + f", seed {cfg['seed']}, exact degrees) x dense F={num_feats} "
                            f"{'fp16' if is_f16 else 'fp32'} -> fp32",
                "num_nodes": num_nodes, "nnz": nnz, "feat": num_feats, "tc_blocks_rank0": total_blocks,
                "timed_call": ("voltrix.spmm_weighted(csr_preprocess_weighted handle, ...)" if args.weighted else
                               "voltrix.spmm(*csr_preprocess handle, ...) -- the drop-in operator") + ", output allocation included"
                              + (" -- on the TRANSPOSED matrix (the backward product dB = A^T dC)" if args.backward else ""),
                "tile": tile_desc,
                "sparse_format": fmt,
                "parallelism": f"row-window shards x{world}" + (
                    " (every rank generates its own shard) + RCCL all-gather(B) per step"
                    + (" (overlapped with the previous step's SpMM)" if overlap_timed else " (dependent: gather, then product)")
                    + (f"; exchange: {gather_mode}" + (" (chosen by the warm-up's measurement)" if args.gather == "auto" else "")
                       + (f", {args.slabs} feature slabs pipelined" if slab_pipeline else ""))
                    if distributed else ""),
                "preprocess_ms": preprocess_ms,
                "fp32_in_ms_per_step": fp32_in_ms, "fp32_in_runs_as": fp32_as,
                "rowsum_check_max_rel_err": check_err,
                "cache_state": "warm (steps back to back; B stays in the Infinity Cache when it fits)",
                "hbm_gbs_algorithmic": synth_graphs.algorithmic_bytes(num_nodes, nnz, num_feats, in_bytes)
                / (ms_per_step * 1e-3) / 1e9,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": counters["traffic_bytes"] if counters else None,
                "mfma_busy_frac": counters.get("mfma_busy_frac") if counters else None,
                "l2_hit_frac": counters.get("l2_hit_frac") if counters else None,
                "counters_from": counters.get("source") if counters else why_not,
                "kernel_sources_hash": sources_hash,
                "kernel": kernels + " (HIP events on the launch stream around the operator call)",
                "kernel_ms": kernel_ms, "kernels_ms": kernels_ms, "algorithmic_bytes": alg_bytes,
                "gather_bytes": gather_bytes, "gather_gbs": gather_bytes / (kernel_ms * 1e-3) / 1e9,
                "note": "gather-bound: B rows are served by L2 / Infinity Cache (~8.6-19 TB/s row-gather ceilings), "
                        "see DESIGN.md section 5",
                "gather_ceiling": gather_ceiling(gather_bytes, num_cols * num_feats * in_bytes, kernel_ms),
                "gather_model": gather_model(gather_bytes, counters.get("l2_hit_frac") if counters else None,
                                                 num_cols * num_feats * in_bytes, kernel_ms),
            },
        }
        line["config"].update(extras)
        if world == 1 and not args.no_cpu_baseline:
            vb = vendor_baseline(local_indptr, local_indices, num_nodes, num_feats, device)
            if "ms" in vb:
                vb["speedup_of_this_work"] = vb["ms"] / ms_per_step
            vb["rocsparse"] = rocsparse_baseline(local_indptr, local_indices, num_nodes, gathered, ms_per_step)
            line["vendor_gpu_baseline"] = vb
            line["cpu_baseline"] = cpu_baseline(local_indptr, local_indices, num_nodes, num_nodes, num_feats)
        if result_fd is not None:
            sys.stdout.flush()
            os.write(result_fd, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
