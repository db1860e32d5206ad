for m in 131072 0; do
CHAOREC_PF_CLS_MIN_ITEMS=$m timeout 900 python3 bench.py --dataset config5_shard --dim 128 --steps 6 --warmup 2 --no-hbm-regime --no-cpu-baseline --no-models > gpurun_out/r05_s_shard_$m.json 2> gpurun_out/r05_s_shard_$m.err; echo "rc=$?"
python3 - <<EOF
import json
d=json.loads(open("gpurun_out/r05_s_shard_$m.json").read().strip().splitlines()[-1])
r=d["roofline_scoring"]; print($m, "score ms", d["config"]["gene_ranklist_ms"], "frac", r["frac"], "sweep_only", r["sweep_only_frac"], r["prefilter"])
EOF
done
CHAOREC_PF_CLS_MIN_ITEMS=131072 REPS=1 TIMES=1 timeout 600 python3 tools/score_case.py 262144 262144 128 2>&1 | tail -1
