#!/bin/bash
# rocprofv3 PMC passes around bench.py (one counter group per pass, --kernel-trace + --pmc only), per-kernel summary and
# the per-step figures bench.py reports as roofline.traffic / mfma_busy_frac / l2_hit_frac (profiles/traffic.json).
#   usage (on the GPU box): harness/pmc_bench.sh <outdir under gpurun_out> <bench.py args...>
# FETCH_SIZE / WRITE_SIZE are in KB; gfx950 counts a 128-B request of a 16-B-per-lane stream as 64 B: FETCH_SIZE x 2
# (MI355X_MICROARCH.md, HBM).  These fabric-side counters include Infinity-Cache hits.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pass$i.json 2> $OUT/pass$i.err
  echo "pass $i ($CTRS) exit=$?"
done
python3 - <<PY
import csv, glob, collections, json
KEEP = ('spmm_tc16_kernel', 'spmm_tc16_pair_kernel', 'spmm_stream_kernel', 'spmm_panel_kernel', 'spmm_fused_kernel', 'add_inplace_f32_kernel', 'combine_partials_kernel', 'FillFunctor<float>')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pass*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if any(k in r['Kernel_Name'] for k in KEEP):
            agg[r['Kernel_Name'][:96]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pass1/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if any(k in r['Kernel_Name'] for k in KEEP):
            dur[r['Kernel_Name'][:96]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
mean = lambda v: sum(v) / len(v) if v else 0.0
lines, step = [], collections.defaultdict(float)
for k, d in agg.items():
    lines.append(f"{k}   calls={len(dur[k])} mean_ms_serialised={mean(dur[k]):.4f}")
    for c, v in sorted(d.items()):
        lines.append(f"  {c:32s} n={len(v):3d} mean={mean(v):.6g}")
    step['fetch_kb'] += mean(d.get('FETCH_SIZE', []))
    step['write_kb'] += mean(d.get('WRITE_SIZE', []))
    step['hit'] += mean(d.get('TCC_HIT_sum', []))
    step['miss'] += mean(d.get('TCC_MISS_sum', []))
    step['mfma_busy'] += mean(d.get('SQ_VALU_MFMA_BUSY_CYCLES', []))
    step['serial_ms'] += mean(dur[k])
entry = {
    "traffic_bytes": int((2 * step['fetch_kb'] + step['write_kb']) * 1024),
    "fetch_kb_sum": step['fetch_kb'], "write_kb_sum": step['write_kb'], "fetch_correction": 2.0,
    "l2_hit_frac": step['hit'] / max(1.0, step['hit'] + step['miss']),
    "mfma_busy_cycles_per_step": step['mfma_busy'],
    "kernels_serialised_ms": step['serial_ms'],
}
lines.append("per step (sum over the step's kernels): " + json.dumps(entry))
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(entry, open("$OUT/traffic_entry.json", "w"), indent=1)
print("\n".join(lines))
PY
