# round 6, batch g: the three-bytes split (CHAOREC_X3_SPLIT=1) -- GEMM users' tests, model steps, one-stream profiles
cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_real_data.py tests/test_gpu_sparse_family.py tests/test_gpu_round3.py tests/test_gpu_feature_adam.py -q -m gpu 2>&1 | tail -12
for m in MMGCN FREEDOM; do
  timeout 600 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'])"
done
echo "== profiles"; timeout 1500 python tools/collect_model_profiles.py 2>&1 | tail -5
