timeout 900 python3 tools/score_mid_sizes.py 2>&1 | grep 'D=' > gpurun_out/r05_zzz_mid_sizes.txt; cat gpurun_out/r05_zzz_mid_sizes.txt | head -3
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_zzz_gpu_tests.log; cat gpurun_out/r05_zzz_gpu_tests.log
timeout 1500 python3 bench.py > gpurun_out/r05_zzz_bench_line.json 2> gpurun_out/r05_zzz_bench_err.log; echo "bench rc=$?"
python3 - <<EOF
import json
d=json.loads(open("gpurun_out/r05_zzz_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"])
print("sports scoring", d["roofline_scoring"]["frac"], d["config"]["gene_ranklist_ms"], d["config"].get("gene_ranklist_ms_cold"))
for k in ("hbm_regime","config5_whole_on_one_gpu"):
    h=d.get(k)
    if h: print(k, {kk: h[kk] for kk in h if kk in ("value","ms_per_step","value_performed")}, "scoring", h.get("roofline_scoring",{}).get("frac"), h.get("roofline_scoring",{}).get("sweep_only_frac"))
print("models", {k: v.get("ms_per_step") for k,v in d.get("models",{}).items()})
EOF
