cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_real_data.py tests/test_gpu_round3.py -q -m gpu 2>&1 | grep -v "^  /\|Warning\|warnings.warn\|^$" | tail -25 | cut -c1-400
echo "== gemm shapes (WS)"; timeout 300 python tools/gemm_wide_bench.py mmgcn 2>&1 | tail -7
echo "== gemm shapes (plain)"; CHAOREC_X3_WS=0 timeout 300 python tools/gemm_wide_bench.py mmgcn 2>&1 | tail -7
for m in MMGCN FREEDOM; do
  timeout 600 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'])"
done
CHAOREC_X3_WS=0 timeout 600 python bench.py --model MMGCN --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MMGCN plain GEMMs', d['ms_per_step'])"
