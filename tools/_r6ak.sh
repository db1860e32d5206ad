cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_dist2.py tests/test_gpu_config5_full.py -q -m gpu -x 2>&1 | tail -2
CHAOREC_BENCH_DETAIL=gpurun_out/r06_al_detail.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-models 2>/dev/null | tail -1 | cut -c1-100
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_al_detail.json'))
for k in ('hbm_regime','config5_whole_on_one_gpu'):
    print(k, d[k]['ms_per_step'])
    for l in d[k]['roofline']['light_step_launches']['launches']:
        print('   ', l['launch'][:60].ljust(60), round(l['us'],1), round(l['frac'],4))
PY
