timeout 900 python3 bench.py --dataset config5_shard --dim 128 --steps 6 --warmup 2 --no-hbm-regime --no-cpu-baseline --no-models > gpurun_out/r05_x_shard.json 2> gpurun_out/r05_x_shard.err; echo "rc=$?"
python3 - <<EOF
import json
d=json.loads(open("gpurun_out/r05_x_shard.json").read().strip().splitlines()[-1])
r=d["roofline_scoring"]; print("score ms", d["config"]["gene_ranklist_ms"], "frac", r["frac"], "sweep_only", r["sweep_only_frac"]); print(r["kernel"]); print(r["sweep_alone"])
EOF
