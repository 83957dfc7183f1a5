#!/usr/bin/env python3
"""Per-step counter summary of a harness/pmc_bench.sh output directory for the kernels the bench line actually timed.

pmc_bench.sh sums every kept kernel of a pass; on a fresh box the first pass also runs the tuner's sweep (dozens of other
tile instantiations), which must not be counted.  This reads the bench line of the last pass (tile, schedule, format, steps),
keeps the kernels of THAT configuration, takes each counter's mean over the last `steps` launches of its pass (the timed
steps) and writes <dir>/summary_step.txt + <dir>/traffic_entry.json (the `runs` entry of profiles/traffic.json).

    python harness/pmc_summarize.py gpurun_out/pmc_products_f512 [--into profiles/traffic.json --source profiles/r03/...]
"""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--into", default=None)
    ap.add_argument("--source", default=None)
    args = ap.parse_args()
    lines = [json.loads(l) for p in sorted(glob.glob(os.path.join(args.dir, "pass*.json"))) for l in open(p) if l.startswith("{")]
    bench = lines[-1]
    cfg, steps = bench["config"], bench["steps"]
    tile = cfg["tile"]
    fmt = cfg["sparse_format"]["format"]
    two_level = fmt.startswith("two-level")
    pair = "two units per wave" in (tile.get("schedule") or "")
    stream = "stream of stages" in (tile.get("schedule") or "")
    tile_sig = f"SpmmTile<{tile['fs']}, {tile['depth']}, {tile['waves']},"
    wanted = {("spmm_tc16_pair_kernel" if pair else ("spmm_stream_kernel" if stream else "spmm_tc16_kernel<")): tile_sig, "combine_partials_kernel": "", "combine_panel_partials_kernel": "" if two_level else None,
              "spmm_panel_kernel": "" if two_level else None, "FillFunctor<float>": "" if two_level else None,
              # the one-launch form only when IT is the timed form (bench.py also times it once "for the record" beside the pair:
              # those launches are not steps)
              "spmm_fused_kernel": "" if ("one launch" in fmt or "one-launch" in fmt or "fused" in fmt) and not two_level else None}

    # wide operands: the window and panel kernels are launched once per group of column slabs (config.tile.launches_per_step)
    per_step = int(tile.get("launches_per_step", 1))
    launches = {"spmm_tc16_pair_kernel": per_step, "spmm_tc16_kernel": per_step, "spmm_panel_kernel": per_step,
                "spmm_stream_kernel": per_step}

    def role(name):
        for key, sig in wanted.items():
            if sig is not None and key in name and sig in name:
                return key.rstrip("<")
        return None

    per = collections.defaultdict(lambda: collections.defaultdict(list))      # role -> counter -> values in launch order
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(args.dir, "pass*", "*counter_collection.csv"))):
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = role(r["Kernel_Name"])
            if k:
                by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, counters in by.items():
            for c, v in counters.items():
                per[k][c] += v[-steps * launches.get(k, 1):]
    for f in sorted(glob.glob(os.path.join(args.dir, "pass1", "*kernel_trace.csv"))):
        rows = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = role(r["Kernel_Name"])
            if k:
                rows[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        for k, v in rows.items():
            dur[k] = v[-steps * launches.get(k, 1):]
    mean = lambda v: sum(v) / len(v) if v else 0.0                             # noqa: E731
    out, step = [], collections.defaultdict(float)
    for k in sorted(per):
        n = launches.get(k, 1)
        out.append(f"{k}: last {steps} steps per pass ({n} launch(es) per step), serialised by the profiler "
                   f"{n * mean(dur[k]):.4f} ms per step")
        for c, v in sorted(per[k].items()):
            out.append(f"  {c:30s} n={len(v):3d} mean per launch={mean(v):.6g}")
        d = per[k]
        step["fetch_kb"] += n * mean(d.get("FETCH_SIZE", []))
        step["write_kb"] += n * mean(d.get("WRITE_SIZE", []))
        step["hit"] += n * mean(d.get("TCC_HIT_sum", []))
        step["miss"] += n * mean(d.get("TCC_MISS_sum", []))
        step["mfma"] += n * mean(d.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        step["busy"] += n * mean(d.get("GRBM_GUI_ACTIVE", []))
        step["ms"] += n * mean(dur[k])
    alg = bench["roofline"]["algorithmic_bytes"]
    traffic = int((2 * step["fetch_kb"] + step["write_kb"]) * 1024)
    entry = {"traffic_bytes": traffic, "fetch_kb_sum": step["fetch_kb"], "write_kb_sum": step["write_kb"],
             "fetch_correction": 2.0, "traffic_over_algorithmic": traffic / alg,
             "l2_hit_frac": step["hit"] / max(1.0, step["hit"] + step["miss"]),
             # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE / 8 XCDs = kernel cycles
             "mfma_busy_frac": step["mfma"] / max(1.0, step["busy"] / 8 * 1024),
             "kernels_serialised_ms": step["ms"], "bench_ms_per_step_under_profiler": bench["ms_per_step"],
             "fabric_TBps_at_serialised_time": traffic / max(1e-9, step["ms"] * 1e-3) / 1e12,
             "source": args.source or args.dir,
             # the kernel sources the counters were measured on (bench.py replays the entry only while they match the tree)
             "sources_hash": bench["roofline"].get("kernel_sources_hash")}
    sched = tile.get("sched")     # the schedule's number (bench lines since round 6); older lines: from its description
    if sched is None:
        sched = 5 if pair else (6 if 'stream' in tile['schedule'] else ('4' if 'unit table' in tile['schedule'] else ('0' if 'natural' in tile['schedule'] else '?')))
    key = (f"{cfg['workload'].split(':')[0]}|F{cfg['feat']}|{bench['dtype']}|{'two-level' if two_level else 'window'}|"
           f"{tile['fs']},{tile['depth']},{tile['waves']}|sched{sched}")
    out.append(f"per step [{key}]: " + json.dumps(entry))
    open(os.path.join(args.dir, "summary_step.txt"), "w").write("\n".join(out) + "\n")
    json.dump({key: entry}, open(os.path.join(args.dir, "traffic_entry.json"), "w"), indent=1)
    print("\n".join(out))
    if args.into:
        data = json.load(open(args.into))
        data.setdefault("runs", {})[key] = entry
        json.dump(data, open(args.into, "w"), indent=1)


if __name__ == "__main__":
    main()
