#!/bin/bash
# usage: harness/experiments/pmc.sh <outdir-name> <bench args...>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/pass$i.json 2> $OUT/pass$i.err
  echo "pass $i ($CTRS) exit=$?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pass*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if any(k in r['Kernel_Name'] for k in ('spmm_tc16', 'spmm_panel', 'add_inplace')):
            agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
with open("$OUT/summary.txt", "w") as out:
    for k, d in agg.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            out.write(f"  {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open("$OUT/summary.txt").read())
PY
