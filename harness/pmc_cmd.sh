#!/bin/bash
# rocprofv3 PMC passes around ANY python script of this repository (one counter group per pass, --kernel-trace + --pmc only:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), per-kernel means for the kernels whose name contains KEEP.
#   usage (on the GPU box): KEEP=spmm_stream harness/pmc_cmd.sh <outdir under gpurun_out> <script.py> [args...]
# FETCH_SIZE / WRITE_SIZE are in KB; gfx950 counts a 128-B request of a 16-B-per-lane stream as 64 B: FETCH_SIZE x 2.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
KEEP=${KEEP:-spmm}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pass$i -o p -- python3 $SCRIPT "$@" > $OUT/pass$i.out 2> $OUT/pass$i.err
  echo "pass $i ($CTRS) exit=$?"
done
KEEP="$KEEP" OUT="$OUT" python3 - <<'PY'
import csv, glob, collections, os
keep, out = os.environ["KEEP"], os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pass*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if keep in r['Kernel_Name']:
            agg[r['Kernel_Name'][:110]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/pass1/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if keep in r['Kernel_Name']:
            dur[r['Kernel_Name'][:110]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
mean = lambda v: sum(v) / len(v) if v else 0.0
lines = []
for k, d in agg.items():
    lines.append(f"{k}   calls={len(dur[k])} mean_ms={mean(dur[k]):.4f} min_ms={min(dur[k]) if dur[k] else 0:.4f}")
    for c, v in sorted(d.items()):
        lines.append(f"  {c:32s} n={len(v):3d} mean={mean(v):.6g}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
