#!/usr/bin/env python3
"""Merge the graph-statistics BUCKET entries of a tuner store (tuned.json written on the GPU box, harness/collect_tuned.sh or
``VOLTRIX_TUNED_STORE=... harness/eval_set.py``) into the shipped defaults (voltrix/jit_kernels/tuned_defaults.json).  Exact-key
entries (they name a matrix tag) stay out.  Every float16 bucket also ships as its bfloat16 twin (same kernels, same traffic).
    python harness/merge_tuned.py gpurun_out/r05/tuned/tuned.json [--dry-run]"""
import argparse
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULTS = os.path.join(os.path.dirname(HERE), "voltrix-spmm_amd", "voltrix", "jit_kernels", "tuned_defaults.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("store")
    ap.add_argument("--dry-run", action="store_true")
    args = ap.parse_args()
    new = json.load(open(args.store))
    shipped = json.load(open(DEFAULTS))
    added = changed = 0
    for key, point in new.items():
        if not key.startswith("spmm_kernel@bucket|"):
            continue
        twins = [(key, point)]
        if "'dtype': 'torch.float16'" in key:
            twins.append((key.replace("'dtype': 'torch.float16'", "'dtype': 'torch.bfloat16'"), dict(point, BF16=1)))
        for k, p in twins:
            if k not in shipped:
                added += 1
            elif shipped[k] != p:
                changed += 1
            shipped[k] = p
    print(f"{added} new bucket entries, {changed} changed, {len(shipped) - 1} in all")
    if not args.dry_run:
        json.dump(shipped, open(DEFAULTS, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
