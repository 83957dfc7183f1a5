set -u
mkdir -p gpurun_out/r06
timeout -k 10 300 python harness/experiments/exp_reorder_eval.py protein_like,reddit_like,fraud_yelp_rsr_like 128 clusters > gpurun_out/r06/reorder_eval_clusters_dense.jsonl 2> gpurun_out/r06/reorder_eval_clusters_dense.err
python - <<PY
import json
for l in open("gpurun_out/r06/reorder_eval_clusters_dense.jsonl"):
    d = json.loads(l); c = d["clusters"]
    print(d["graph"], round(d["natural_ms"], 4), round(d["shuffled_ms"], 4), round(c.get("reordered_ms", 0), 4), c.get("vs_natural"), round(c.get("local", 0), 3), c.get("wall_ms"), c.get("error"))
PY
timeout -k 10 700 python harness/experiments/exp_reorder_eval.py > gpurun_out/r06/reorder_eval_auto_all.jsonl 2> gpurun_out/r06/reorder_eval_auto_all.err
python - <<PY
import json
for l in open("gpurun_out/r06/reorder_eval_auto_all.jsonl"):
    d = json.loads(l); c = d["auto"]
    print(d["graph"], round(d["natural_ms"], 4), round(d["shuffled_ms"], 4), c.get("picked"), round(c.get("reordered_ms", 0), 4), c.get("vs_natural"), c.get("vs_shuffled"), round(c.get("local", 0), 3), c.get("wall_ms"), c.get("estimates"), c.get("error"))
PY
