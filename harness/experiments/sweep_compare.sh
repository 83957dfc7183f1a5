set -u
O=gpurun_out/r06/sweep_compare; mkdir -p $O
for G in copurchase_half amazon0505_like amazon0601_like com_amazon_like web_berkstan_like dd_like ppi_like; do
  for M in shipped swept; do
    VOLTRIX_PRINT_AUTO_TUNE=1 timeout -k 10 300 python tests/tuner_heldout_worker.py $M $G 128 $O/store_${G}_$M.json $O/${G}_$M.json > $O/${G}_$M.log 2>&1 || tail -3 $O/${G}_$M.log
  done
  python - <<PY
import json
a, b = json.load(open("$O/${G}_shipped.json")), json.load(open("$O/${G}_swept.json"))
print("$G", "shipped", round(a["step_ms"], 4), [(p["FS"], p["DEPTH"], p["WAVES"], p["SCHED"]) for p in a["points"]], a["tuner"]["bucket_hits"], "swept", round(b["step_ms"], 4), [(p["FS"], p["DEPTH"], p["WAVES"], p["SCHED"]) for p in b["points"]], b["tuner"]["timed_candidates"])
PY
done
grep "has time" $O/copurchase_half_swept.log | sed 's/.*tuned keys//' | sort -t' ' -k1 | head -80
