#!/bin/bash
set -u
O=gpurun_out/r06/csr; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csr_path.py -m gpu -q -x > $O/pytest_csr.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_csr.log
timeout -k 10 900 python harness/eval_set.py --datasets DD,com-amazon,amazon0601,amazon0505,ppi,web-BerkStan,Yeast,YeastH --methods rocSPARSE-best,Voltrix,Voltrix-fp16 --check --output_file $O/results_csr.csv --jsonl $O/eval_csr.jsonl > $O/eval_csr.log 2>&1; echo "eval rc=$?"
python - <<PY
import json
for l in open("$O/eval_csr.jsonl"):
    d = json.loads(l)
    if d["method"].startswith("Voltrix"):
        print(d["dataset"], d["feat"], d["method"], round(d["steady_ms"], 4), d.get("path"), d.get("calc_diff_vs_hipsparse"))
PY
