# round 6, batch ab: column-striped long rows of the list launches
cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-300
for t in 8192; do
  echo "== CHAOREC_ROWLIST_STRIPE_T=$t default bench"
  CHAOREC_ROWLIST_STRIPE_T=$t CHAOREC_BENCH_DETAIL=gpurun_out/r06_ab_detail_stripe$t.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_ab_detail_stripe8192.json'))
for k in ('hbm_regime','config5_whole_on_one_gpu'):
    print(k, d[k]['ms_per_step'])
    for l in d[k]['roofline']['light_step_launches']['launches']:
        print('   ', l['launch'][:60].ljust(60), round(l['us'],1), round(l['frac'],4))
PY
