#!/bin/bash
# usage: harness/experiments/pmc2.sh <outdir-name> <run_list args...>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/harness/experiments/run_list.py "$@" > /dev/null 2> $OUT/pass$i.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pass*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if 'spmm_' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pass1/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if 'spmm_' in r['Kernel_Name']: dur[r['Kernel_Name'][:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k, d in agg.items():
    print(k, 'ms=%.3f' % (sum(dur[k])/max(1,len(dur[k]))), ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
