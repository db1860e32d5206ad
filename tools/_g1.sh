python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_v_gpu_tests.log; cat gpurun_out/r05_v_gpu_tests.log
timeout 1200 python3 bench.py --dataset config5 --dim 128 --steps 3 --warmup 1 --no-hbm-regime --no-cpu-baseline --no-models > gpurun_out/r05_v_full.json 2> gpurun_out/r05_v_full.err; echo "rc=$?"
python3 - <<EOF
import json
d=json.loads(open("gpurun_out/r05_v_full.json").read().strip().splitlines()[-1])
r=d["roofline_scoring"]; print("score ms", d["config"]["gene_ranklist_ms"], "frac", r["frac"], "sweep_only", r["sweep_only_frac"], r["prefilter"]); print("ms_per_step", d["ms_per_step"])
EOF
