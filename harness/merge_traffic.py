#!/usr/bin/env python3
"""Bring the per-step PMC summaries of harness/final_measure.sh pmc (gpurun_out/<out>/pmc_<workload>/) into the tracked tree:
the summary text to profiles/<round>/pmc_<name>_f128.txt and the entry to profiles/traffic.json (entries measured on other kernel
sources of the same round -- a stale sources_hash with a source under profiles/<round>/ -- are dropped).
    python harness/merge_traffic.py gpurun_out/r05/final r05"""
import glob
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
NAMES = {"headline": "pmc_reddit_f128_two_level_pairs.txt"}


def main():
    out, rnd = sys.argv[1], sys.argv[2]
    path = os.path.join(REPO, "profiles", "traffic.json")
    data = json.load(open(path))
    runs = data["runs"]
    fresh = None
    for d in sorted(glob.glob(os.path.join(out, "pmc_*", ""))):
        name = os.path.basename(os.path.dirname(d))[4:]
        (key, entry), = json.load(open(os.path.join(d, "traffic_entry.json"))).items()
        target = NAMES.get(name) or (f"pmc_{name}.txt" if re.search(r"_f\d+$", name) else f"pmc_{name}_f128.txt")
        entry["source"] = f"profiles/{rnd}/{target}"
        runs[key] = entry
        fresh = entry["sources_hash"]
        shutil.copy(os.path.join(d, "summary_step.txt"), os.path.join(REPO, "profiles", rnd, target))
        print(key, round(entry["traffic_bytes"] / 1e6), "MB", round(entry["traffic_over_algorithmic"], 2), "x  L2",
              round(entry["l2_hit_frac"], 3), " ms", round(entry["kernels_serialised_ms"], 4), entry["sources_hash"])
    for k in [k for k, v in runs.items() if str(v.get("source", "")).startswith(f"profiles/{rnd}/") and v.get("sources_hash") != fresh]:
        print("dropped (other kernel sources):", k)
        del runs[k]
    json.dump(data, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
