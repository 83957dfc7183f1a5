"""Round 6: can the two kernels of the two-level step gain from DISJOINT CU sets (CU-masked streams,
hipExtStreamCreateWithCUMask) instead of sharing every CU?  Headline graph, F = 128 fp16.
  1. each kernel ALONE on 256 / 192 / 128 / 64 CUs: if its time goes with 1 / CUs it is bound per CU (gather requests, matrix
     cores), and a split takes max(t_w / x, t_p / (1 - x)) >= t_w + t_p > the shared-CU pair;
  2. the pair on disjoint sets at three splits around the predicted optimum;
  3. the shipped pair (all CUs shared).
Mask bits: the pattern (bit // 8) % 4 < k keeps k / 4 of the CUs of every XCD whether the driver numbers CUs XCD-interleaved or
XCD-contiguous.  Kill criterion: < 5 % gain -> record and stop.
    python harness/experiments/exp_cu_mask.py"""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "voltrix-spmm_amd")
sys.path[:0] = [REPO, PKG]
os.environ.setdefault("VOLTRIX_CACHE_DIR", os.path.join(PKG, ".jit_cache"))
os.environ["VOLTRIX_HYBRID"] = "1"

import torch  # noqa: E402

import synth_graphs  # noqa: E402
import voltrix  # noqa: E402
from voltrix import hybrid  # noqa: E402
from voltrix.jit_kernels.spmm import spmm_kernel  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(keep):
    """A stream restricted to the CUs i with keep(i) (256 bits)."""
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if keep(i):
            words[i // 32] |= 1 << (i % 32)
    handle = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(handle.value), sum(1 for i in range(256) if keep(i))


def main():
    dev = torch.device("cuda", 0)
    indptr, indices, cfg = synth_graphs.generate("reddit_like", device=dev)
    n, e, f = indptr.numel() - 1, indices.numel(), 128
    feat = torch.randn(n, f, device=dev).half()
    handle = voltrix.csr_preprocess_device(indptr, indices, n)
    handle[1].hash_tag = "bench/reddit_like/s1.0/r0of1"          # the shipped / persisted tile choice of the headline
    two = voltrix.two_level_of(handle[1])
    for _ in range(3):
        voltrix.spmm(*handle, num_nodes=n, num_edges=e, feat=feat)
    out = torch.zeros(n, f, device=dev)

    def window(stream):
        with torch.cuda.stream(stream):
            pending = spmm_kernel(two.blk_offsets, two.hspa_packed, two.hind, num_nodes=n, num_edges=two.plan.num_resid_edges,
                                  embedding_dim=f, input=feat, output=out, atomic_out=True, beside_panel=True, defer_combine=True,
                                  xcd_ptr=two.window_xcd_ptr)
        return pending

    def panel(stream):
        hybrid.launch_panel(two.plan, feat, out, accumulate=2, stream=stream.cuda_stream, defer_combine=True)

    def time_on(fn_list, iters=10):
        """fn_list: [(fn, stream)]; all enqueued per iteration, events on each stream; -> per-stream ms, wall ms of the step."""
        torch.cuda.synchronize()
        per, both = [0.0] * len(fn_list), 0.0
        for it in range(iters + 2):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in fn_list]
            torch.cuda.synchronize()
            for (fn, st), (a, b) in zip(fn_list, evs):
                a.record(st)
                fn(st)
                b.record(st)
            torch.cuda.synchronize()
            if it >= 2:
                for k, (a, b) in enumerate(evs):
                    per[k] += a.elapsed_time(b) / iters
                both += max(evs[k][0].elapsed_time(evs[j][1]) for k in range(len(evs)) for j in range(len(evs))) / iters
        return [round(p, 4) for p in per], round(both, 4)

    full = torch.cuda.current_stream()
    side = hybrid.side_stream(dev)
    line = {"graph": "reddit_like", "F": f}
    alone = {}
    for k in (4, 3, 2, 1):
        st, cus = masked_stream(lambda i, k=k: (i // 8) % 4 < k)
        alone[cus] = {"window_ms": time_on([(window, st)])[0][0], "panel_ms": time_on([(panel, st)])[0][0]}
    line["alone_by_cus"] = alone
    line["alone_unmasked"] = {"window_ms": time_on([(window, full)])[0][0], "panel_ms": time_on([(panel, side)])[0][0]}
    line["pair_shared_cus"] = dict(zip(("per_stream_ms", "step_ms"), time_on([(window, full), (panel, side)])))
    splits = {}
    for k in (3, 2, 1):                       # window kernel on k / 4 of every XCD's CUs, panel kernel on the rest
        sw, cw = masked_stream(lambda i, k=k: (i // 8) % 4 < k)
        sp, cp = masked_stream(lambda i, k=k: (i // 8) % 4 >= k)
        splits[f"window {cw} CUs | panel {cp} CUs"] = dict(zip(("per_stream_ms", "step_ms"), time_on([(window, sw), (panel, sp)])))
    line["pair_disjoint_cus"] = splits
    t_w, t_p = line["alone_unmasked"]["window_ms"], line["alone_unmasked"]["panel_ms"]
    line["model"] = {"sum_alone_ms": round(t_w + t_p, 4),
                     "note": "per-CU-bound kernels on disjoint sets: max(t_w / x, t_p / (1 - x)) >= t_w + t_p (x = t_w / (t_w + t_p))"}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
