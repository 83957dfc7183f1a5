"""The fp32 -> scaled-fp16 operand cast of the reference-protocol path (fp32 features in, voltrix/spmm/spmm.py::_operand):
its three launches timed alone, warm and after a 512-MiB cache flush, against the bytes they move (amax: 4 B per element
read; cast: 4 B read + 2 B written).
    python harness/experiments/exp_cast.py"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "voltrix-spmm_amd")]

import torch  # noqa: E402

from voltrix import capi  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for feat in (128, 256, 512, 1024):
        n = 232965
        x = torch.randn(n, feat, device=dev)
        out = torch.empty(n, feat, dtype=torch.float16, device=dev)
        scale = torch.empty(2, dtype=torch.float32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        run = lambda: capi.launch_cast_f32_f16_scaled(x, out, scale, s)  # noqa: E731
        for _ in range(3):
            run()
        res = {}
        for label, cold in (("warm", False), ("cold", True)):
            times = []
            for _ in range(15):
                if cold:
                    flush.zero_()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run()
                b.record()
                b.synchronize()
                times.append(a.elapsed_time(b))
            res[label] = sorted(times)[len(times) // 2]
        moved = n * feat * (4 + 4 + 2)
        print(json.dumps({"F": feat, "elements": n * feat, "bytes_moved_MB": moved / 1e6, "warm_ms": round(res["warm"], 4),
                          "cold_ms": round(res["cold"], 4), "cold_TBps": round(moved / res["cold"] / 1e9, 2),
                          "warm_TBps": round(moved / res["warm"] / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
