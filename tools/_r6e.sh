# round 6, batch e: gated SpMM compaction -- tests, the bench line, D = 64 scoring PMC
cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_fused_step.py tests/test_gpu_config5.py tests/test_gpu_config5_full.py tests/test_gpu_round5.py -q -m gpu -x 2>&1 | tail -15
echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_e_bench.out 2> gpurun_out/r06_e_bench.err; tail -c 3500 gpurun_out/r06_e_bench.out; cp bench_detail.json gpurun_out/r06_e_bench_detail.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_e_bench_detail.json'))
for k in ('hbm_regime','config5_whole_on_one_gpu'):
    s=d.get(k,{})
    print(k, 'ms_per_step', s.get('ms_per_step'), 'build', s.get('host_build_seconds'))
    for l in ((s.get('roofline') or {}).get('light_step_launches') or {}).get('launches',[]):
        print('   %-80s %9.1f us  frac %.3f'%(l['launch'][:80], l['us'], l['frac']))
for n,m in d.get('models',{}).items(): print(n, m.get('ms_per_step'))
PY
echo "== PMC: the D = 64 scoring kernels of the steady sports call"
EPOCH_APART=1 PMC_TIMEOUT=500 PMC_PASSES=0,1,2,3,7 timeout 2700 python tools/pmc_kernels.py score_sweep score_select -- python3 $GRAFT_REPO_ROOT/tools/score_profile.py 600 > gpurun_out/r06_e_score_pmc_d64.txt 2>&1
tail -70 gpurun_out/r06_e_score_pmc_d64.txt
