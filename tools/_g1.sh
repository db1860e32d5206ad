timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_config5.py -q -m gpu 2>&1 | tail -3
CHAOREC_PF_CLS_MIN_ITEMS=1 timeout 600 python3 tools/score_sorted_probe.py 4096 200000 2>&1 | grep 'sorted'
timeout 900 python3 bench.py --dataset config5_shard --dim 128 --steps 6 --warmup 2 --no-hbm-regime --no-cpu-baseline --no-models 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_scoring']; print('shard score ms', d['config']['gene_ranklist_ms'], 'frac', r['frac'], r['kernel'][:40], r['prefilter'])"
