"""Per-XCD / per-CU finish times of the window kernel ("report per-CU time histogram /
tail" on the power-law stress graph).  A diagnostic build of the kernel (VOLTRIX_DIAG bit 3: results unchanged) lets every
wave record where it ran (XCC id, HW id) and when (s_memrealtime, 100 MHz); the script launches the natural order, the
round-1 balance schedule, the unit table and the paired unit table through the C-ABI and prints, per schedule, the launch
duration, the finish time of every XCD, percentiles of the CUs' last-wave finish times and the tail (how long the slowest
CU runs after the median CU has gone idle).

    python harness/experiments/exp_tail_histogram.py build
    python harness/experiments/exp_tail_histogram.py run [workload] [scale] [feat]
    python harness/experiments/exp_tail_histogram.py pair [workload] [scale] [feat]    the two-level step: the stamped window kernel
                                                    on the residual (two units per wave, atomic epilogue) BESIDE the panel kernel
"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "voltrix-spmm_amd")
sys.path[:0] = [REPO, PKG]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG, ".jit_cache"))
SO = os.path.join(HERE, "build", "tail_stamps_f16.so")


def build():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DVOLTRIX_DIAG=8", "-DVOLTRIX_EXPERIMENTAL", f"-I{PKG}/voltrix/include", f"-I{REPO}/include", f"-I{PKG}/csrc",
                           os.path.join(PKG, "csrc", "capi_spmm_f16.hip"), "-o", SO])


def run():
    import torch

    import synth_graphs
    import voltrix
    from voltrix import capi
    from voltrix.jit_kernels.spmm import ORDER_CHUNKS, PAIR_UNIT_FACTOR
    from voltrix.schedule import default_max_stages, unit_table

    pair_mode = sys.argv[1] == "pair"
    workload = sys.argv[2] if len(sys.argv) > 2 else ("reddit_like" if pair_mode else "powerlaw_4m")
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    feat_dim = int(sys.argv[4]) if len(sys.argv) > 4 else synth_graphs.CONFIGS[workload]["feat"]
    dev = torch.device("cuda")
    os.environ["VOLTRIX_HYBRID"] = "1" if pair_mode else "0"
    indptr, indices, _ = synth_graphs.generate(workload, device=dev, scale=scale)
    n, nnz = indptr.numel() - 1, indices.numel()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    two = None
    if pair_mode:
        two = voltrix.two_level_of(handle[1])        # the side-car csr_preprocess_device attached (VOLTRIX_HYBRID=1)
        assert two is not None
        handle = two.residual
        side = torch.cuda.Stream()
    del indptr, indices
    blk = handle[0]
    nst = ((blk[1:] - blk[:-1]) + 3) // 4
    print(json.dumps({"workload": workload, "scale": scale, "N": n, "nnz": nnz, "F": feat_dim, "tc_blocks": int(blk[-1]),
                      "stages_per_window": {"median": float(nst.float().median()), "p99": float(nst.float().quantile(0.99)),
                                            "max": int(nst.max())}}), flush=True)
    feat = torch.randn(n, feat_dim, device=dev).half()
    out = torch.empty(n, feat_dim, device=dev)
    lib = ctypes.CDLL(SO)
    fn = lib.voltrix_launch_spmm_f16_sched
    fn.restype = None
    stream = torch.cuda.current_stream().cuda_stream
    tile = (128, 3, 4)
    slabs = (feat_dim + tile[0] - 1) // tile[0]
    windows = (n + 15) // 16
    order = torch.empty(windows, dtype=torch.int32, device=dev)
    capi.launch_window_order(blk, n, order, stream, ORDER_CHUNKS[3])
    tb = unit_table(blk, n)
    tb_p = unit_table(blk, n, max(8, int(PAIR_UNIT_FACTOR * default_max_stages(blk, n) / 1.5)))
    cases = {"natural order": dict(), f"balance schedule (chunk {ORDER_CHUNKS[3]})": dict(order=order),
             "unit table": dict(table=tb), "unit table, two units per wave": dict(table=tb_p, upw=2)}
    if pair_mode:
        cases = {"residual alone: unit table, two units per wave": dict(table=tb_p, upw=2),
                 "residual beside the panel kernel (the two-level step)": dict(table=tb_p, upw=2, panel=True)}
    for name, c in cases.items():
        table = c.get("table")
        units_per_xcd = table.max_units_per_xcd if table is not None else (windows + 7) // 8
        per_xcd = ((units_per_xcd + 1) // 2 if c.get("upw") == 2 else units_per_xcd) * slabs
        grid = (per_xcd + tile[2] - 1) // tile[2] * 8
        stamps = torch.full((grid * tile[2], 4), -1, dtype=torch.int32, device=dev)
        partials = torch.empty(max(1, table.num_slots if table is not None else 1) * 16 * feat_dim, device=dev)

        def launch(stamp_ptr):
            rc = ctypes.c_int(-1)
            if c.get("panel"):   # zero fill, then the panel kernel on the side stream, as voltrix.hybrid does
                out.zero_()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    assert capi.launch_spmm_panel(two.plan, feat.data_ptr(), out.data_ptr(), feat_dim, 2, False, (128, 3, 1), 0,
                                                  side.cuda_stream) == 0
            fn(capi._ptr(handle[0]), capi._ptr(handle[1]), capi._ptr(handle[2]), ctypes.c_int(n), ctypes.c_int(nnz),
               ctypes.c_int(feat_dim), capi._ptr(feat), capi._ptr(out), ctypes.c_int(tile[0]), ctypes.c_int(tile[1]),
               ctypes.c_int(tile[2]), ctypes.c_void_p(c["order"].data_ptr() if "order" in c else 0), ctypes.c_void_p(0),
               ctypes.c_int(1 if pair_mode else 0), ctypes.c_void_p(table.units.data_ptr() if table is not None else 0),
               ctypes.c_void_p(table.unit_ptr.data_ptr() if table is not None else 0),
               ctypes.c_int(table.max_units_per_xcd if table is not None else 0), capi._ptr(partials),
               ctypes.c_void_p(stamp_ptr), ctypes.c_int(c.get("upw", 1)), ctypes.c_void_p(stream), ctypes.byref(rc))
            assert rc.value == 0, rc.value
            if c.get("panel"):
                torch.cuda.current_stream().wait_stream(side)

        for _ in range(2):
            launch(stamps.data_ptr())
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stamps.fill_(-1)
        s.record()
        launch(stamps.data_ptr())
        e.record()
        e.synchronize()
        st = stamps[stamps[:, 0] >= 0].cpu().long()
        t0 = int(st[:, 2].min())
        end = ((st[:, 3] - t0) & 0xFFFFFFFF).double() / 100.0          # microseconds since the first wave started
        xcc = st[:, 0] & 0xF
        hw = st[:, 1]
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)   # XCC, SE, SH, CU
        cu_ids, inv = torch.unique(cu, return_inverse=True)
        cu_end = torch.zeros(cu_ids.numel(), dtype=torch.double).scatter_reduce_(0, inv, end, reduce="amax")
        xcd_end = [round(float(end[xcc == x].max()), 1) if bool((xcc == x).any()) else None for x in range(8)]
        q = torch.quantile(cu_end, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.double)).tolist()
        print(json.dumps({"schedule": name, "launch_ms": round(s.elapsed_time(e), 3), "waves": int(st.shape[0]),
                          "cus_seen": int(cu_ids.numel()), "xcd_finish_us": xcd_end,
                          "cu_finish_us": {"min": round(q[0], 1), "p10": round(q[1], 1), "median": round(q[2], 1),
                                           "p90": round(q[3], 1), "max": round(q[4], 1)},
                          "tail_after_median_cu_frac": round((q[4] - q[2]) / q[4], 4),
                          "cu_finish_histogram_10_bins": torch.histc(cu_end.float(), bins=10, min=0, max=float(q[4])).int().tolist()}),
              flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()   # "run" or "pair"
