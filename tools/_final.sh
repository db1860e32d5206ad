# the round's closing run: the whole GPU suite, then the default bench line (both kept under profiles/)
timeout 2300 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r05_zzz_gpu_tests.log
cut -c1-300 gpurun_out/r05_zzz_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 700 python bench.py > gpurun_out/r05_zzz_default_bench_line.json 2> gpurun_out/bench.err < /dev/null
echo rc=$?
python3 - <<'P'
import json
l = json.loads(open("gpurun_out/r05_zzz_default_bench_line.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["scoring"]["frac"], l["hbm_regime"]["roofline_scoring"]["frac"],
      l["config5_whole_on_one_gpu"]["roofline_scoring"]["frac"], {k: (v["ms_per_step"], v["value"]) for k, v in l["models"].items()})
P
